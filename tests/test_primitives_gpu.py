"""Device unit tests of the hand-written scan and radix sort (semantic_dsp_map_amd/csrc/primitives.hip)."""
import numpy as np
import pytest

from semantic_dsp_map_amd import binding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2047, 2048, 2049, 100000, 4194304 + 17])
def test_exclusive_scan(n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 9, n, dtype=np.uint32)
    got = binding.test_scan(a)
    want = np.concatenate([[0], np.cumsum(a[:-1], dtype=np.uint64)]).astype(np.uint32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,nbits", [(1, 8), (64, 8), (2048, 9), (2049, 16), (300000, 25), (1 << 20, 31), (777777, 32)])
def test_radix_sort_pairs_is_stable(n, nbits):
    rng = np.random.default_rng(n + nbits)
    # few distinct keys -> long equal-key runs, which is what exercises stability
    hi = min((1 << nbits) - 1, 5000 if n > 10000 else 7)
    keys = rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    if nbits == 32 or nbits == 31:
        keys = (keys.astype(np.uint64) * ((1 << nbits) // (hi + 1))).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = binding.test_sort_pairs(keys, vals, nbits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ko, keys[order])
    assert np.array_equal(vo, vals[order])
