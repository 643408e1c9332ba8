"""The C++20 mirror of bvh::v2 (include/bvh/v2/) over the C-ABI: it must compile with plain g++ (no HIP headers) and,
on a GPU, reproduce the known answer of the reference's test/simple_example.cpp."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "simple_example_amd.cpp")


def _compile(out, src=SRC):
    from bvh_amd import build
    build.build()
    lib = os.path.join(ROOT, "bvh_amd", "lib")
    cmd = ["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src,
           "-L", lib, "-lbvh_amd", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_cpp_mirror_compiles_with_gxx(tmp_path):
    _compile(str(tmp_path / "circles_2d_amd"), os.path.join(ROOT, "tests", "cpp", "circles_2d_amd.cpp"))     # Node<T, 2>
    _compile(str(tmp_path / "serialize_amd"), os.path.join(ROOT, "tests", "cpp", "serialize_amd.cpp"))       # streams
    _compile(str(tmp_path / "replicate_amd"), os.path.join(ROOT, "tests", "cpp", "replicate_amd.cpp"))       # multi-GPU mirror
    _compile(str(tmp_path / "template_knobs_amd"), os.path.join(ROOT, "tests", "cpp", "template_knobs_amd.cpp"))   # BinCount, Index bits
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


@pytest.mark.gpu
def test_cpp_serialize_round_trip(tmp_path, orc):
    """test/serialize.cpp's flow through the mirror; the file it writes is the reference's stream for the same build."""
    import numpy as np
    exe = _compile(str(tmp_path / "serialize_amd"), os.path.join(ROOT, "tests", "cpp", "serialize_amd.cpp"))
    path = str(tmp_path / "bvh.bin")
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "The deserialized BVH is the same as the original one" in r.stdout
    tris = []
    for i in range(40):
        x = float(i)
        tris.append([x + 1, -1, 1, x + 1, 1, 1, x, 1, 1])
        tris.append([x + 1, -1, 1, x, -1, 1, x, 1, 1])
    bb, cc = orc.prep_tris(np.array(tris, dtype=np.float32))
    assert open(path, "rb").read() == orc.build(bb, cc, quality=2).serialize()


@pytest.mark.gpu
def test_cpp_2d_circles_match_reference(tmp_path, orc):
    """Bvh<Node<float, 2>> through the mirror: node count and every hit record equal the reference's."""
    import numpy as np
    exe = _compile(str(tmp_path / "circles_2d_amd"), os.path.join(ROOT, "tests", "cpp", "circles_2d_amd.cpp"))
    n = 4096
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    s = np.uint64(12345)
    vals = []
    with np.errstate(over="ignore"):
        for _ in range(3 * n):                                # the program's LCG
            s = s * np.uint64(6364136223846793005) + np.uint64(1442695040888963407)
            vals.append(np.float32(float(int(s) >> 40) * (1.0 / 16777216.0)))
    v = np.array(vals, dtype=np.float32).reshape(n, 3)
    circ = np.stack([v[:, 0], v[:, 1], np.float32(0.001) + np.float32(0.004) * v[:, 2]], axis=1).astype(np.float32)
    bb, cc = orc.sphere_bboxes(circ)
    ref = orc.build(bb, cc, quality=2)
    rays = np.array([[-0.1, i / 1000, 1, 0, 0, 100] for i in range(1000)], dtype=np.float32)
    rays[:, 1] = (np.arange(1000, dtype=np.float32) / np.float32(1000))
    want = ref.intersect_sphere(circ[ref.prim_ids().astype(np.int64)], rays, 0, 1)
    head = lines[0].split()
    assert int(head[1]) == ref.node_count and int(head[3]) == int((want["prim"] != 0xFFFFFFFF).sum())
    for line, w in zip(lines[1:], want):
        p, t, u = line.split()
        assert int(p) == int(w["prim"]) and np.float32(t) == w["t"] and np.float32(u) == w["u"]


def _repack(stream: bytes, index_bits: int, count_bits: int) -> bytes:
    """The reference's stream for the same tree held in Node<float, 3, index_bits, count_bits> (node.h:90-94, bvh.h:221-229: counts,
    ids and the index word as Index::Type; index.h:73-77: first_id << PrimCountBits | prim_count)."""
    import numpy as np
    from conftest import parse_stream
    nodes, ids = parse_stream(stream)
    it = np.dtype("<u8" if index_bits == 64 else "<u4")
    first, count = (nodes["index"] >> 4).astype(np.uint64), (nodes["index"] & 15).astype(np.uint64)
    out = np.zeros(len(nodes), dtype=np.dtype([("bounds", "<f4", (6,)), ("index", it)]))
    out["bounds"] = nodes["bounds"]
    out["index"] = ((first << np.uint64(count_bits)) | count).astype(it)
    return np.array([len(nodes), len(ids)], dtype=it).tobytes() + out.tobytes() + ids.astype(it).tobytes()


def test_index_repacking_rule_matches_reference_streams(orc):
    """What the mirror does to a Node with non-default Index parameters (re-pack first_id / prim_count of the same tree) is what the
    reference's own instantiations serialize (tests/golden/template_knobs.npz, made by Node<float, 3, 32, 2> / Node<float, 3, 64, 6>)."""
    import oracle
    from conftest import load_golden
    g, gk = load_golden("soup2k"), load_golden("template_knobs")
    bb, cc = g["bboxes"], g["centers"]
    assert gk["soup2k_count2bits_sweep"].tobytes() == _repack(orc.build(bb, cc, builder=oracle.BUILDER_SWEEP, max_leaf=3).serialize(), 32, 2)
    high = orc.build(bb, cc, builder=oracle.BUILDER_DEFAULT_SERIAL, quality=oracle.QUALITY_HIGH)
    assert gk["soup2k_index64_high"].tobytes() == _repack(high.serialize(), 64, 6)
    first = int(high.nodes()["index"][0]) >> 4
    assert gk["soup2k_index64_high_sub"].tobytes() == _repack(high.extract(first).serialize(), 64, 6)


@pytest.mark.gpu
def test_cpp_template_knobs_match_reference(tmp_path):
    """BinnedSahBuilder<Node, 4 | 16 | 32> and Node<T, Dim, IndexBits, PrimCountBits> through the mirror (tests/cpp/template_knobs_amd.cpp)
    on the soup2k fixture's input: every stream equals the one the reference's own template instantiation wrote (template_knobs.npz)."""
    import numpy as np
    from conftest import load_golden
    exe = _compile(str(tmp_path / "template_knobs_amd"), os.path.join(ROOT, "tests", "cpp", "template_knobs_amd.cpp"))
    g, gk = load_golden("soup2k"), load_golden("template_knobs")
    bb, cc = g["bboxes"], g["centers"]
    with open(tmp_path / "input.bin", "wb") as f:
        f.write(np.ascontiguousarray(bb, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(cc, dtype=np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path / "input.bin"), str(len(bb)), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
    read = lambda name: open(tmp_path / name, "rb").read()
    for bins in (4, 16, 32):
        assert read(f"bins{bins}.bin") == gk[f"soup2k_bins{bins}"].tobytes(), bins
        assert read(f"bins{bins}_leaf2to5.bin") == gk[f"soup2k_bins{bins}_leaf2to5"].tobytes(), bins
    for name in ("count2bits_sweep", "index64_high", "index64_high_sub"):
        assert read(f"{name}.bin") == gk[f"soup2k_{name}"].tobytes(), name


def test_cpp_traverse_top_down_with_a_steering_inner_fn(tmp_path):
    """Bvh::traverse_top_down<IsAnyHit>(start, stack, leaf_fn, inner_fn) of the mirror (reference bvh.h:125-157): inner_fn returns
    {visit left, visit right, right first}; a host-side utility over the mirror's nodes (no device involved). The expected visit orders
    come from a restatement of the reference's loop below."""
    exe = _compile(str(tmp_path / "top_down_amd"), os.path.join(ROOT, "tests", "cpp", "top_down_amd.cpp"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    # nodes: (x0, x1, first, count); count 0 = inner with children first, first + 1
    nodes = [(0, 5, 1, 0), (0, 3, 3, 0), (3, 5, 5, 0), (0, 1, 0, 1), (1, 3, 1, 2), (3, 4, 3, 1), (4, 5, 4, 1)]

    def walk(inner_fn, any_hit=False, hit=lambda b, e: False):
        out, stack = [], [0]
        while stack:                                          # bvh.h:128-156
            top = stack.pop()
            dead = False
            while nodes[top][3] == 0:
                l, rgt = nodes[top][2], nodes[top][2] + 1
                hl, hr, swap = inner_fn(nodes[l], nodes[rgt])
                if hl:
                    near = l
                    if hr:
                        far = rgt
                        if swap:
                            near, far = far, near
                        stack.append(far)
                    top = near
                elif hr:
                    top = rgt
                else:
                    dead = True
                    break
            if dead:
                continue
            b, e = nodes[top][2], nodes[top][2] + nodes[top][3]
            out.append(f"[{b},{e})")
            if any_hit and hit(b, e):
                break
        return " ".join(out)
    want = [walk(lambda l, r: (True, True, False)), walk(lambda l, r: (True, True, True)), walk(lambda l, r: (l[0] < 2.5, r[0] < 2.5, False)),
            walk(lambda l, r: (True, True, False), True, lambda b, e: b <= 3 < e)]
    got = [line.strip() for line in r.stdout.strip().splitlines()]
    assert got == want, (got, want)
    assert want[0] == "[0,1) [1,3) [3,4) [4,5)" and want[1] == "[4,5) [3,4) [1,3) [0,1)" and want[3] == "[0,1) [1,3) [3,4)"
