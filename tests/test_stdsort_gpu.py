"""The wave-parallel std::sort restatement the GPU runs on merged reads that hold a position twice (csrc/lps_graph.hip wave_std_sort: introsort
loop with every partition done by 64 lanes, stable rank count for the final insertion-sort phase) against the real libstdc++ std::sort, on
(key, payload) pairs compared by key only - bit-exact including the order left among equal keys (src/phase/PhasingGraph.cpp:854)."""
import ctypes as C
import os

import numpy as np
import pytest

from lps import hip

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def oracle():
    lib = C.CDLL(os.path.join(HERE, "..", "oracle", "liblps_oracle.so"))
    lib.oracle_std_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    return lib


def check(rows):
    rows = [np.ascontiguousarray(r, dtype=np.int32) for r in rows]
    start = np.zeros(len(rows) + 1, np.int64); start[1:] = np.cumsum([r.size for r in rows])
    keys = np.concatenate(rows) if rows else np.zeros(0, np.int32)
    pay = (np.arange(keys.size) % 251).astype(np.uint8)
    k1, p1, k2, p2 = keys.copy(), pay.copy(), keys.copy(), pay.copy()
    assert hip.load().lps_debug_std_sort_gpu(0, k1.ctypes.data, p1.ctypes.data, start.ctypes.data, len(rows)) == 0
    ora = oracle(); unstable = 0
    for r in range(len(rows)):
        a, b = int(start[r]), int(start[r + 1])
        ora.oracle_std_sort(k2[a:b].ctypes.data, p2[a:b].ctypes.data, b - a)
        assert np.array_equal(k1[a:b], k2[a:b]) and np.array_equal(p1[a:b], p2[a:b]), (r, b - a, rows[r][:24])
        unstable += not np.array_equal(p2[a:b], pay[a:b][np.argsort(rows[r], kind="stable")])
    return unstable


def test_merged_runs_with_duplicates():
    rng = np.random.default_rng(21); rows = []
    for trial in range(6000):
        runs = []; lo = 0
        for r in range(int(rng.integers(2, 6))):
            n = int(rng.integers(3, 200 if trial % 7 == 0 else 60))
            a = np.sort(rng.choice(np.arange(lo, lo + 4 * n), n, replace=False))
            if runs and rng.random() < 0.8:
                k = int(rng.integers(1, min(6, n, runs[-1].size) + 1)); a[:k] = runs[-1][-k:]
            runs.append(a); lo = int(a[-1]) - int(rng.integers(0, 3))
        rows.append(np.concatenate(runs))
    assert check(rows) > 600


def test_adversarial_and_edge_shapes():
    rng = np.random.default_rng(22); rows = []
    for n in (0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 66, 100, 127, 128, 129, 500, 1000, 1023, 1024, 1025, 3000):
        rows += [rng.integers(0, max(1, n // 3), n), rng.integers(0, 3, n), np.arange(n), np.arange(n)[::-1], np.zeros(n),
                 np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]])]
    for n in (64, 256, 1024, 2048):                                      # median-of-three killer: reaches the heapsort fallback
        a = np.zeros(n, np.int64); k = n // 2
        for i in range(1, k + 1):
            if i & 1:
                a[i - 1] = i; a[i] = k + i
            a[k + i - 1] = 2 * i
        rows.append(a)
    check(rows)
