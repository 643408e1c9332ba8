import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def orc():
    """The CPU restatement of the reference (oracle/bvh_oracle.cpp); compiled on demand."""
    import oracle
    return oracle.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The unmodified reference compiled in place (oracle/_ref); skipped where it cannot exist."""
    import oracle
    lib = oracle.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libbvh_ref.so not available (no /root/reference on this machine)")
    return lib


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


MODES = [("binned", 2, 0), ("sweep", 3, 0),
         ("serial_low", 0, 0), ("serial_med", 0, 1), ("serial_high", 0, 2),
         ("parallel_low", 1, 0), ("parallel_med", 1, 1), ("parallel_high", 1, 2)]


def parse_stream(buf: bytes, double=False):
    """Decodes Bvh::serialize output (reference bvh.h:221-229) into (nodes, prim_ids)."""
    import oracle
    idx = np.dtype("<u8" if double else "<u4")
    node = oracle.NODED if double else oracle.NODEF
    hdr = np.frombuffer(buf, dtype=idx, count=2)
    nn, npr = int(hdr[0]), int(hdr[1])
    off = 2 * idx.itemsize
    nodes = np.frombuffer(buf, dtype=node, count=nn, offset=off)
    ids = np.frombuffer(buf, dtype=idx, count=npr, offset=off + nn * node.itemsize)
    assert off + nn * node.itemsize + npr * idx.itemsize == len(buf)
    return nodes, ids.astype(np.uint64)
