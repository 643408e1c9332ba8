"""-m gpu: the order-defining sorts: exact libstdc++ std::sort emulation and the stable radix sort."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _keysets(dtype):
    rng = np.random.default_rng(1)
    sets = []
    for n in (0, 1, 2, 15, 16, 17, 18, 33, 100, 1000, 2048, 4095, 4096, 4097, 65536, 300_000):     # <= 4096: the one-launch LDS kernel
        sets.append(rng.random(n).astype(dtype))                                  # distinct-ish
        sets.append(rng.integers(0, 4, n).astype(dtype))                          # 4 distinct keys: ties dominate
        sets.append((rng.integers(0, 1000, n) / 8).astype(dtype))                 # many ties
        sets.append(np.sort(rng.random(n)).astype(dtype))                         # pre-sorted
        sets.append(np.sort(rng.random(n))[::-1].astype(dtype).copy())            # reverse-sorted
        sets.append(np.zeros(n, dtype))                                           # all equal
        k = rng.random(n).astype(dtype) - 0.5
        if n:
            k[::5] = 0.0
            k[::7] = -0.0
        sets.append(k)                                                            # signed zeros + negatives
    # organ-pipe and sawtooth patterns stress the median-of-3 / depth budget
    for n in (100_000, 4096, 3000):
        sets.append(np.concatenate([np.arange(n // 2), np.arange(n // 2)[::-1]]).astype(dtype))
        sets.append((np.arange(n) % 17).astype(dtype))
    # median-of-3 killer sequences (Musser): drive introsort into its heap-sort fallback (depth 0)
    for n in (4096, 1024, 60_000):
        k = n // 2
        a = np.empty(n)
        a[0:2 * k:2] = np.arange(1, k + 1)
        a[1:2 * k:2] = np.arange(k + 1, 2 * k + 1)
        sets.append(a.astype(dtype))
    return sets


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_std_sort_emulation_matches_libstdcxx(orc, dtype):
    import bvh_amd
    for keys in _keysets(dtype):
        want = orc.std_sort_ids(keys)
        got = bvh_amd.std_sort_ids(keys).cpu().numpy().astype(np.uint32) if len(keys) else np.empty(0, np.uint32)
        assert (got == want).all(), (len(keys), dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_std_sort_with_nan_keys_stays_a_permutation(dtype):
    """std::sort is undefined for NaN keys; whatever order comes out, every id appears exactly once (both the LDS kernel and the
    partition-replay + radix path)."""
    import bvh_amd
    rng = np.random.default_rng(5)
    for n in (17, 1000, 4096, 20_000):
        keys = rng.random(n).astype(dtype)
        keys[rng.integers(0, n, max(1, n // 10))] = np.nan
        got = bvh_amd.std_sort_ids(keys).cpu().numpy().astype(np.int64)
        assert (np.sort(got) == np.arange(n)).all(), (n, dtype)


def test_radix_sort_is_stable():
    import bvh_amd
    rng = np.random.default_rng(2)
    for n, bits in ((1, 8), (255, 8), (4096, 12), (4097, 12), (1_000_000, 12), (500_000, 32), (70_000, 20)):
        keys = rng.integers(0, 1 << min(bits, 31), n, dtype=np.int64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k, v = bvh_amd.radix_sort_pairs(keys.view(np.int32), vals.view(np.int32), bits)
        order = np.argsort(keys, kind="stable")
        assert (k.cpu().numpy().view(np.uint32) == keys[order]).all()
        assert (v.cpu().numpy().view(np.uint32) == vals[order]).all()
