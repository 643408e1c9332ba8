"""The C++20 mirror of bvh::v2 (include/bvh/v2/) over the C-ABI: it must compile with plain g++ (no HIP headers) and,
on a GPU, reproduce the known answer of the reference's test/simple_example.cpp."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "simple_example_amd.cpp")


def _compile(out):
    from bvh_amd import build
    build.build()
    lib = os.path.join(ROOT, "bvh_amd", "lib")
    cmd = ["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", lib, "-lbvh_amd", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_cpp_mirror_compiles_with_gxx(tmp_path):
    exe = _compile(str(tmp_path / "simple_example_amd"))
    import torch
    if not torch.cuda.is_available():                         # no GPU: the program must fail loudly, not fall back
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode != 0 and "no ROCm-capable device" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_cpp_simple_example_known_answer(tmp_path):
    exe = _compile(str(tmp_path / "simple_example_amd"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "primitive: 1" in out and "distance: 1" in out and "barycentric coords.: -0, 0.5" in out
    assert "nodes: 1, prim_ids: 1 0" in out                   # serial-High stream of test/serialize.cpp: ids [1, 0]
