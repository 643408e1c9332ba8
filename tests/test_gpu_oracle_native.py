"""-m gpu (it has to run on the GPU box's HOST): SURVEY.md Appendix A.1 pins the reference build `g++ -O3 -march=native -ffp-contract=off`;
the shipped oracle/_ref/libbvh_ref.so says `-mavx2 -mfma` instead so that it runs anywhere (oracle/ref_harness.cpp:13-18). This closes the gap
(VERDICT r4 Weak 9): oracle/_ref/libbvh_ref_native.so is the same reference compiled with what `-march=native` expands to ON THAT HOST
(oracle/gpu_box_native_flags.txt, captured there: znver3 + AVX-512), and every builder mode x scene, the reinsertion optimizer, refit
and all four traversal modes must give the same bytes from both builds. Skipped where the library is absent or the CPU lacks AVX-512."""
import numpy as np
import pytest

import oracle
from bvh_amd import synth
from conftest import MODES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def both():
    ref = oracle.load_ref()
    if ref is None:
        pytest.skip("no oracle/_ref/libbvh_ref.so in this tree")
    if not oracle.cpu_has_avx512():
        pytest.skip("this host's CPU lacks the AVX-512 subsets the GPU box's -march=native build was compiled for")
    nat = oracle.load_ref_native()
    if nat is None:
        pytest.skip("no oracle/_ref/libbvh_ref_native.so in this tree (make -C oracle ref_native needs /root/reference)")
    return ref, nat


def _scene(name):
    if name == "soup":
        return synth.soup(60000, seed=5, jitter=0.02)
    if name == "terrain":
        return synth.terrain(60000)
    if name == "sponza":
        return synth.sponza_proxy(60000)
    if name == "clusters":
        return synth.clusters(60000)
    raise KeyError(name)


@pytest.mark.parametrize("scene", ["soup", "terrain", "sponza", "clusters"])
def test_native_build_of_the_reference_equals_the_shipped_build(both, scene):
    ref, nat = both
    tris = _scene(scene)
    bb, cc = ref.prep_tris(tris)
    bb2, cc2 = nat.prep_tris(tris)
    assert bb.tobytes() == bb2.tobytes() and cc.tobytes() == cc2.tobytes()
    lo, hi = synth.scene_bounds(tris)
    rays = synth.rays_closest(200_000, lo, hi)
    srays = synth.rays_shadow(200_000, lo, hi)
    for mode, builder, quality in MODES:                      # 8 modes: binned, sweep, serial / parallel x Low / Medium / High
        a = ref.build(bb, cc, builder=builder, quality=quality, threads=8)
        b = nat.build(bb, cc, builder=builder, quality=quality, threads=8)
        assert a.serialize() == b.serialize(), (scene, mode)
    pa, pb = ref.precompute_tris(tris, a.prim_ids()), nat.precompute_tris(tris, b.prim_ids())
    assert pa.tobytes() == pb.tobytes()
    for any_hit in (0, 1):
        for robust in (0, 1):
            rr = srays if any_hit else rays
            ha, ca = a.intersect_tri(pa, rr, any_hit, robust, threads=8, counters=True)
            hb, cb = b.intersect_tri(pb, rr, any_hit, robust, threads=8, counters=True)
            assert ha.tobytes() == hb.tobytes() and (ca == cb).all(), (scene, any_hit, robust)
    a.optimize(3)
    b.optimize(3)
    assert a.serialize() == b.serialize()
    a.refit()
    b.refit()
    assert a.serialize() == b.serialize()


def test_native_build_double_spheres(both):
    ref, nat = both
    sph = synth.spheres(50000, rmin=0.005, rmax=0.02)
    bb, cc = ref.sphere_bboxes(sph)
    bb2, cc2 = nat.sphere_bboxes(sph)
    assert bb.tobytes() == bb2.tobytes() and cc.tobytes() == cc2.tobytes()
    a = ref.build(bb, cc, builder=1, quality=2, threads=8)
    b = nat.build(bb, cc, builder=1, quality=2, threads=8)
    assert a.serialize() == b.serialize()
    perm = a.prim_ids().astype(np.int64)
    lo, hi = synth.scene_bounds(sph)
    rays = synth.rays_closest(100_000, lo, hi, dtype=np.float64)
    for any_hit in (0, 1):
        for robust in (0, 1):
            ha, ca = a.intersect_sphere(sph[perm], rays, any_hit, robust, threads=8, counters=True)
            hb, cb = b.intersect_sphere(sph[perm], rays, any_hit, robust, threads=8, counters=True)
            assert ha.tobytes() == hb.tobytes() and (ca == cb).all()
