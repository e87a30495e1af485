"""CPU test: the oracle's Philox4x32-10 restatement (oracle/philox.py) against Random123's published known-answer vectors, and the
statistical sanity of the index / normal mappings built on it (the GPU tests compare the device streams with this restatement)."""
import numpy as np

from oracle import philox


def test_philox_known_answer_vectors():
    for ctr, key, want in philox.KAT:
        got = philox.philox4x32_10(np.array(ctr, dtype=np.uint64), np.array(key, dtype=np.uint64))
        assert tuple(int(x) for x in got) == want, (ctr, key, [hex(int(x)) for x in got])


def test_philox_is_a_bijection_on_a_sample():
    rs = np.random.RandomState(0)
    ctr = rs.randint(0, 2 ** 32, size=(4096, 4), dtype=np.uint64)
    out = philox.philox4x32_10(ctr, np.array([1, 2], dtype=np.uint64))
    assert len({tuple(r) for r in out.tolist()}) == len(ctr)


def test_streams_are_consistent_prefixes_and_distinct():
    a = philox.raw_stream(1000, 7, 1 << 40)
    assert np.array_equal(a[:333], philox.raw_stream(333, 7, 1 << 40))         # element e does not depend on n
    assert not np.array_equal(a, philox.raw_stream(1000, 7, (1 << 40) + 1))     # next train() counter value: a new stream
    assert not np.array_equal(a, philox.raw_stream(1000, 8, 1 << 40))           # another seed (rank): a new stream


def test_index_and_normal_mappings():
    i = philox.indices(1 << 18, 1000, 5, 1 << 40)
    assert i.min() == 0 and i.max() == 999
    cnt = np.bincount(i, minlength=1000)
    chi2 = ((cnt - i.size / 1000) ** 2 / (i.size / 1000)).sum()
    assert chi2 < 999 + 5 * 44.7
    x = philox.normals(1 << 20, 0.5, 5, 2 << 40).astype(np.float64) / 0.5
    n = x.size
    assert abs(x.mean()) < 5 / np.sqrt(n) and abs(x.var() - 1) < 5 * np.sqrt(2 / n) and abs((x ** 4).mean() - 3) < 5 * np.sqrt(96 / n)
