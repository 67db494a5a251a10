"""pytest configuration: `gpu` marker, paths, and the oracle (test-only checker) loader."""
import ctypes
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_binding import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def corpus():
    """[(name, compressed bytes, expected length, expected sha256, expected bytes or None)]"""
    d = os.path.join(GOLDEN, "decodecorpus")
    manifest = json.load(open(os.path.join(d, "manifest.json")))
    out = []
    for stem in sorted(manifest):
        m = manifest[stem]
        comp = open(os.path.join(d, stem + ".zst"), "rb").read()
        exp = open(os.path.join(d, stem), "rb").read() if m["verbatim"] else None
        out.append((stem, comp, m["length"], m["sha256"], exp))
    return out


@pytest.fixture(scope="session")
def kat():
    return json.load(open(os.path.join(GOLDEN, "kat.json")))


def check_expected(name, got, length, sha, exp):
    assert len(got) == length, f"{name}: length {len(got)} != {length}"
    if exp is not None:
        assert got == exp, f"{name}: bytes differ"
    assert hashlib.sha256(got).hexdigest() == sha, f"{name}: sha256 differs"
