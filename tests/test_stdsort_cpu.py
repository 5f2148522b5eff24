"""The library's restatement of libstdc++'s std::sort (csrc/lps_stdsort.h: introsort loop, median-of-three, unguarded partition, heapsort fallback,
final insertion sort) against the real std::sort, on (key, payload) pairs compared by key only.  The order it leaves among EQUAL keys is what matters:
a merged read that holds a position twice (overlapping supplementary alignments) feeds the reference's fp32 edge sums in that order."""
import ctypes as C
import os

import numpy as np

from lps import hip

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = C.CDLL(os.path.join(HERE, "..", "oracle", "liblps_oracle.so"))
ORACLE.oracle_std_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]


def both(keys):
    keys = np.ascontiguousarray(keys, dtype=np.int32); n = keys.size
    pay = (np.arange(n) % 251).astype(np.uint8)
    k1, p1, k2, p2 = keys.copy(), pay.copy(), keys.copy(), pay.copy()
    hip.load().lps_debug_std_sort(k1.ctypes.data, p1.ctypes.data, n)
    ORACLE.oracle_std_sort(k2.ctypes.data, p2.ctypes.data, n)
    assert np.array_equal(k1, k2) and np.array_equal(p1, p2), (n, keys[:20])
    return not np.array_equal(p1, pay[np.argsort(keys, kind="stable")])


def test_merged_runs_with_duplicates():
    rng = np.random.default_rng(11)
    unstable = 0
    for trial in range(3000):
        runs = []
        lo = 0
        for r in range(int(rng.integers(2, 5))):
            n = int(rng.integers(3, 60))
            a = np.sort(rng.choice(np.arange(lo, lo + 4 * n), n, replace=False))
            if runs and rng.random() < 0.8:                             # share a few positions with the previous run
                k = int(rng.integers(1, min(6, n, runs[-1].size) + 1)); a[:k] = runs[-1][-k:]
            runs.append(a); lo = int(a[-1]) - int(rng.integers(0, 3))
        unstable += both(np.concatenate(runs))
    assert unstable > 300                                               # the cases the stable order would get wrong are exercised


def test_adversarial_and_edge_shapes():
    rng = np.random.default_rng(12)
    for n in (0, 1, 2, 15, 16, 17, 31, 32, 33, 64, 100, 1000, 5000):
        both(rng.integers(0, max(1, n // 3), n))                        # many duplicates
        both(np.arange(n)); both(np.arange(n)[::-1]); both(np.zeros(n))
        both(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]))   # organ pipe
    # median-of-three killer (drives the introsort into its heapsort fallback)
    for n in (64, 256, 1024, 4096):
        a = np.zeros(n, np.int64); k = n // 2
        for i in range(1, k + 1):
            if i & 1:
                a[i - 1] = i; a[i] = k + i
            a[k + i - 1] = 2 * i
        both(a)
