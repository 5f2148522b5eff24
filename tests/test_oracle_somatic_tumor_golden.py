"""CPU tests: the tumor-BAM extraction restatement (row a21) reproduces what the REAL reference logged for its somatic calls
(--somatic-calling-log): 18 integer fields of <prefix>_somatic_var.out exactly, and - through the reference's DenseAlt rule
(SomaticVarCaller.cpp:1160-1204) - the per-site sameCount of <prefix>_densealt_filter.log derived from the +-100 bp difference
windows, at EVERY tumor site."""
import collections

import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi

# (counter column(s) of lps_tumor_extract_result.site, 1-based field of _somatic_var.out)
FIELDS = dict(tumAltCount=([0], 6), caseReadCount=([25, 29], 7), pureH1_1=([26], 9), pureH2_1=([27], 10), pureH3=([28], 11), mixed=([29], 12),
              unTag=([24], 13), tumDepth=([6], 25), tumDel=([7], 28), H1=([16], 35), H2=([17], 36), H1_1=([20], 37), H2_1=([22], 38), H3=([18], 39),
              somH1_1=([35], 52), somH2_1=([37], 53), somH3=([33], 54), somUnTag=([30], 55))


def check_tumor_sites(V, out, what):
    idx = np.searchsorted(V.pos, V.log_pos)
    c = out.site[idx]
    for k, (cols, fld) in FIELDS.items():
        mine = c[:, cols].sum(axis=1)
        ref = V.log_val[:, fld - 1].astype(np.int64)
        assert np.array_equal(mine, ref), f"{what}: {k} differs at {np.nonzero(mine != ref)[0][:5]}"


def dense_alt_same_count(V, out):
    """DenseAlt rule restated (double arithmetic against float thresholds, std::map offset order, early stop at minThr)."""
    ws, wa, wo, _ = out.windows()
    thr1, thr2, mn = float(np.float32(V.dense_thr[0])), float(np.float32(V.dense_thr[1])), int(V.dense_thr[2])
    cnt = collections.defaultdict(lambda: [collections.Counter(), collections.Counter()])
    for s_, a_, o_ in zip(ws, wa, wo):
        cnt[int(s_)][int(a_)][int(o_)] += 1
    same = np.zeros(V.n, np.int64)
    for v, (rc, ac) in cnt.items():
        alt = int(out.site[v, 0]); k = 0
        for off in sorted(ac):
            aa = ac[off]; ra = rc.get(off, 0)
            c1 = aa / alt if alt else float("inf")
            if c1 >= thr1 and aa / (ra + aa) >= thr2:
                k += 1
                if k == mn:
                    break
        same[v] = k
    return same


@pytest.mark.parametrize("name", sorted(fixtures.SOMATIC_FIXTURES))
def test_oracle_tumor_extraction_matches_reference_logs(name):
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    T, R = util.make_tumor_reads(name)
    V, _, _, _ = util.load_golden_somatic(name)
    out = lps_oracle.somatic_extract_tumor(abi.default_params(**over), V, T.ref, R)
    check_tumor_sites(V, out, name)
    same = dense_alt_same_count(V, out)
    didx = np.searchsorted(V.pos, V.dense_pos)
    assert np.array_equal(V.pos[didx], V.dense_pos)
    assert np.array_equal(same[didx], V.dense_cnt), f"{name}: DenseAlt sameCount differs at {np.nonzero(same[didx] != V.dense_cnt)[0][:5]}"
    assert (V.dense_cnt > 0).sum() > 20
