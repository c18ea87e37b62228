"""The container model behind bit-exact pair order (dsm-framework_amd/csrc/setorder.h) against the real
std::unordered_set<unsigned> of this toolchain's libstdc++, and the 30-sample reference golden against the oracle."""
import os
import subprocess

import orc
from goldenlib import server_args_to_kw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_matches_real_unordered_set(tmp_path):
    exe = str(tmp_path / "setorder_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "setorder_check.cpp")], check=True)
    out = subprocess.run([exe, "4000"], check=True, stdout=subprocess.PIPE).stdout
    assert out.startswith(b"ok ")


def test_oracle_matches_reference_with_30_samples(golden):
    m = golden.manifest["sets"]["many30"]
    idx = [orc.Index(golden.fmi("many30", n)) for n in m["names"]]
    for cfg, args in m["server_cfgs"].items():
        for p in m["prefixes"]:
            got, _ = orc.mine(idx, m["names"], [p], fmin=m["fmin"], maxdepth=m["maxdepth"], threads=1, **server_args_to_kw(args))
            assert got == golden.server_out("many30", cfg, p), (cfg, p)
    for ix in idx:
        ix.close()
