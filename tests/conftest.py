import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Golden-parity files first, the full-size runs (minutes of index construction) last: a slow box that runs out of its time limit
# then stops in the size-independent checks, not before the byte-for-byte comparisons with the reference's outputs.
_ORDER = ["test_oracle_golden", "test_setorder", "test_stream_parse", "test_cabi", "test_gpu_parity", "test_server_side", "test_cli_dropin",
          "test_builder", "test_distmat", "test_format_text", "test_dist_exchange", "test_many_samples_gpu", "test_fullsize_gpu"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) - 1  # unknown files: before the full-size file
    items.sort(key=key)  # (stable: the order inside a file is kept)


@pytest.fixture(scope="session")
def golden():
    import goldenlib
    return goldenlib.Golden()
