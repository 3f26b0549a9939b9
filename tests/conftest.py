import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


def pytest_sessionstart(session):
    """The tests exercise the libraries in the tree: make sure they were built from the sources in the tree (a no-op when they are
    up to date; hipcc cross-compiles without a GPU).  A stale libkltgpu.so once cost an afternoon."""
    import shutil
    import subprocess
    if shutil.which("make") is None:
        return
    for sub, target in (("pyfeaturetrack_amd/csrc", "libkltgpu.so"), ("oracle", "libkltoracle.so")):
        d = os.path.join(REPO, sub)
        if subprocess.run(["make", "-q", "-C", d, target], capture_output=True).returncode != 0:
            subprocess.run(["make", "-C", d, "-j4", target], check=False, capture_output=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cfg1():
    return np.load(os.path.join(GOLDEN, "cfg1.npz"))


@pytest.fixture(scope="session")
def synth251():
    return np.load(os.path.join(GOLDEN, "synth251.npz"))


def read_pgm(path):
    """Minimal binary P5 reader (no PIL needed): uint8 [nrows, ncols]."""
    with open(path, "rb") as f:
        data = f.read()
    toks = []
    pos = 0
    while len(toks) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        start = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        toks.append(data[start:pos])
    pos += 1
    assert toks[0] == b"P5" and int(toks[3]) == 255
    w, h = int(toks[1]), int(toks[2])
    return np.frombuffer(data, np.uint8, count=w * h, offset=pos).reshape(h, w).copy()


@pytest.fixture(scope="session")
def img0():
    return read_pgm(os.path.join(GOLDEN, "img0.pgm"))


@pytest.fixture(scope="session")
def img1():
    return read_pgm(os.path.join(GOLDEN, "img1.pgm"))
