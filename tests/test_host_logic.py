"""CPU-only checks: the C-ABI library loads and exports every declared symbol, the host mirror's plumbing,
synthetic generators, and the N>1 path (world_size 2, gloo)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from bvh_amd import synth
from conftest import ROOT, load_golden, parse_stream


def test_library_exports_every_declared_symbol():
    from bvh_amd import _lib, build
    build.build()
    dll = _lib.load()
    header = open(os.path.join(ROOT, "include", "bvh_amd.h")).read()
    declared = set(re.findall(r"BVH_AMD_API[^;]*?\b(bvh\w+)\s*\(", header))
    assert len(declared) > 50
    for name in sorted(declared):
        assert hasattr(dll, name), f"{name} declared in include/bvh_amd.h but not exported"
    assert declared <= set(_lib.exported_symbols())
    assert dll.bvh_amd_version().startswith(b"bvh_amd")


def test_every_function_of_the_reference_c_header_is_exported():
    """tests/golden/c_api_symbols.txt = the 94 names of src/bvh/v2/c_api/bvh.h (make_golden.py --symbols)."""
    from bvh_amd import _lib, build
    build.build()
    dll = _lib.load()
    names = open(os.path.join(ROOT, "tests", "golden", "c_api_symbols.txt")).read().split()
    assert len(names) == 94
    header = open(os.path.join(ROOT, "include", "bvh_amd.h")).read()
    for name in names:
        assert hasattr(dll, name), f"{name}: in the reference's C API, not exported by libbvh_amd.so"
        assert re.search(r"\b" + name + r"\s*\(", header), f"{name}: not declared in include/bvh_amd.h"
    assert os.path.exists(os.path.join(ROOT, "include", "bvh", "v2", "c_api", "bvh.h"))


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bvh_amd
    g = load_golden("cornell")
    with pytest.raises(bvh_amd.BvhAmdError):
        bvh_amd.BinnedSahBuilder.build(g["bboxes"], g["centers"])
    nodes, ids = parse_stream(g["bvh_binned"].tobytes())
    with pytest.raises(bvh_amd.BvhAmdError):
        bvh_amd.Bvh.from_nodes(nodes, ids)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bvh_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_synth_is_deterministic_and_shaped():
    a, b = synth.soup(1000), synth.soup(1000)
    assert a.tobytes() == b.tobytes() and a.shape == (1000, 9) and a.dtype == np.float32
    assert synth.soup(1000, seed=8).tobytes() != a.tobytes()
    # counter-based: a prefix of a longer stream is the same stream
    assert synth.uniform01(3, 10).tobytes() == synth.uniform01(3, 20)[:10].tobytes()
    assert synth.sponza_proxy(262144).shape == (262144, 9)
    t = synth.terrain(20000)
    assert t.shape[1] == 9 and abs(len(t) - 20000) < 500
    lo, hi = synth.scene_bounds(a)
    r = synth.rays_closest(100, lo, hi)
    assert r.shape == (100, 8) and np.allclose(np.linalg.norm(r[:, 3:6], axis=1), 1, atol=1e-6)
    s = synth.rays_shadow(100, lo, hi)
    assert (s[:, 6] == np.float32(1e-4)).all()


def test_shard_ranges_cover_everything():
    from bvh_amd.parallel import shard_range
    for n in (0, 1, 7, 100, 12_500_001):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and a <= b


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from bvh_amd.parallel import broadcast_bytes, broadcast_tensor, shard_range
from conftest import load_golden, parse_stream
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = load_golden("soup2k")
stream = g["bvh_parallel_high"].tobytes()
got = broadcast_bytes(stream if rank == 0 else None, src=0)
assert got == stream
nodes, ids = parse_stream(got)
assert len(nodes) == len(parse_stream(stream)[0])
prims = torch.arange(24, dtype=torch.float32).reshape(2, 12) if rank == 0 else None
prims = broadcast_tensor(prims, src=0, dtype=torch.float32)
assert prims.shape == (2, 12) and float(prims[1, 11]) == 23.0
b, e = shard_range(4096, rank, world)
cover = torch.zeros(4096, dtype=torch.int64); cover[b:e] = 1
dist.all_reduce(cover)
assert int(cover.min()) == 1 and int(cover.max()) == 1
dist.barrier(); dist.destroy_process_group()
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"rank{{rank}}.ok"), "w").write("ok")
"""


def test_two_rank_broadcast_and_sharding_gloo(tmp_path):
    import socket
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    for attempt in range(2):                                  # a fresh free port each time (rendezvous ports can linger)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()      # (stdout of two ranks can interleave)
