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
def _restatement():
    import oracle
    return oracle.load_oracle()


@pytest.fixture(scope="session")
def _compiled_reference():
    """oracle/_ref where it is in the tree — and then it must load (oracle.gpu_checker raises instead of falling back)."""
    import oracle
    lib = oracle.gpu_checker()
    return lib if lib.prefix == "ref" else None


class _PreferReference:
    """The compiled, unmodified reference (oracle/_ref) behind the restatement's Python interface; the one entry point only the
    restatement has (std_sort_ids: libstdc++'s std::sort on a key array) stays with it."""

    def __init__(self, ref, restatement):
        self._ref, self._restatement = ref, restatement

    def __getattr__(self, name):
        return getattr(self._restatement if name == "std_sort_ids" else self._ref, name)


@pytest.fixture
def restatement(_restatement):
    """Always the restatement: for what the compiled reference's harness cannot do (it traces with the SmallStack<Index, 64> of the
    reference's examples, so trees deeper than 64 levels need the restatement's growing stack)."""
    return _restatement


@pytest.fixture
def orc(request, _restatement, _compiled_reference):
    """The CPU checker. CPU tests: the restatement of the reference (oracle/bvh_oracle.cpp; it is what those tests pin against the
    golden vectors). `-m gpu` tests: the compiled, unmodified reference wherever oracle/_ref loads (it does on the GPU box: the .so
    travels with the snapshot) — the restatement only where it cannot exist."""
    if request.node.get_closest_marker("gpu") is not None and _compiled_reference is not None:
        return _PreferReference(_compiled_reference, _restatement)
    return _restatement


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
