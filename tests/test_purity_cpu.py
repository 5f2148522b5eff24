"""cli/cli_purity.h (the tumor purity estimator of `somatic_haplotag`, restating src/somatic_haplotag/TumorPurityEstimator.cpp) on seeded inputs: read-count
histograms with one, two and three modes, ties, tiny inputs that make the estimator give up.  Expected purities, reports and error texts
(tests/golden/purity_cases.json) were produced by the round-2 implementation whose reports tests/test_cli_somatic_gpu.py compares with the reference's
as text; this test lets the estimator be edited without a GPU."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "purity_cases.json")


def make_case(k):
    rng = np.random.default_rng(1000 + k)
    kind = k % 8
    n = int(rng.integers(1, 3000)) if kind != 7 else int(rng.integers(1, 6))
    if kind == 0: counts = rng.poisson(25, n)
    elif kind == 1: counts = np.concatenate([rng.poisson(4, n // 3), rng.poisson(30, n - n // 3)])
    elif kind == 2: counts = np.concatenate([rng.poisson(3, n // 2), rng.poisson(18, n // 4), rng.poisson(50, n - n // 2 - n // 4)])
    elif kind == 3: counts = rng.integers(0, 8, n)
    elif kind == 4: counts = np.concatenate([rng.poisson(10, n // 2), rng.poisson(12, n - n // 2)]) * int(rng.integers(1, 4))
    elif kind == 5: counts = np.concatenate([rng.poisson(6, 4 * n // 5), rng.poisson(60, n - 4 * n // 5)])
    elif kind == 6: counts = np.concatenate([rng.integers(0, 1500, n // 10 + 1), rng.poisson(40, n)])
    else: counts = rng.integers(0, 40, n)
    n = len(counts)
    mode = k % 3
    if mode == 0: ratios = np.clip(rng.normal(0.6, 0.08, n), 0.5, 1.0)
    elif mode == 1: ratios = np.round(np.clip(rng.normal(0.75, 0.15, n), 0.5, 1.0), 2)
    else: ratios = rng.choice([0.5, 0.55, 0.6, 0.9, 1.0], n)
    return f"{n} {n + k} 1 2 3 4 5\n" + "\n".join(f"{x:.17g} {int(y)}" for x, y in zip(ratios, counts)) + "\n"


def run_case(exe, k, workdir):
    out = os.path.join(workdir, "p_purity.out")
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([exe, os.path.join(workdir, "p")], input=make_case(k), capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    report = open(out).read() if os.path.exists(out) else None
    return dict(purity=r.stdout.strip(), stderr=r.stderr, report_sha256=hashlib.sha256(report.encode()).hexdigest() if report is not None else None,
                threshold=[ln.split(": ")[1] for ln in report.splitlines() if "DYNAMIC_THR" in ln][0] if report else None)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("purity")
    out = str(d / "purity_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", os.path.join(HERE, "purity_harness.cpp"), "-o", out, "-ldl", "-lz", "-lpthread"])
    return out


def test_purity_estimator_matches_the_pinned_cases(exe, tmp_path):
    gold = json.load(open(GOLD))
    assert len(gold["cases"]) >= 48
    kinds = set()
    for k, want in enumerate(gold["cases"]):
        got = run_case(exe, k, str(tmp_path))
        assert got == want, (k, got, want)
        kinds.add((want["report_sha256"] is None, want["threshold"] not in (None, "0")))
    assert kinds == {(True, False), (False, False), (False, True)}      # gave up / no valley / a valley threshold: all three ways out are covered


if __name__ == "__main__":        # regenerate: python tests/test_purity_cpu.py <harness built from the implementation to pin>
    import sys, tempfile
    with tempfile.TemporaryDirectory() as d:
        json.dump(dict(note="purity / report / messages per seeded case of tests/test_purity_cpu.py::make_case", cases=[run_case(sys.argv[1], k, d) for k in range(64)]),
                  open(GOLD, "w"), indent=0)
