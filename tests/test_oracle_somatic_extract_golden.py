"""CPU tests: the normal-BAM extraction restatement (row a20) reproduces the per-site values the REAL reference logged
(--somatic-calling-log, <prefix>_somatic_var.out) for its somatic calls: depth, deletion count, ALT count, H1/H2 read counts
in the normal BAM exactly; VAF-type ratios recomputed from the integers within float print precision."""
import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi

SC = abi.SC


def check_normal_sites(V, counters, what):
    idx = np.searchsorted(V.pos, V.log_pos)
    assert np.array_equal(V.pos[idx], V.log_pos)
    L = lambda k: V.log_val[:, util.LOG[k] - 1]
    c = counters[idx]
    assert np.array_equal(c[:, SC["DEPTH"]], L("norDepth").astype(np.int64)), what + ": normal depth"
    assert np.array_equal(c[:, SC["DEL"]], L("norDel").astype(np.int64)), what + ": normal deletion count"
    assert np.array_equal(c[:, SC["ALT"]], L("norAltCount").astype(np.int64)), what + ": normal alt count"
    assert np.array_equal(c[:, SC["READHP_H1"]], L("norH1").astype(np.int64)), what + ": H1 reads in normal BAM"
    assert np.array_equal(c[:, SC["READHP_H2"]], L("norH2").astype(np.int64)), what + ": H2 reads in normal BAM"
    # calculateBaseCommonInfo (SomaticVarCaller.cpp:13-40): VAF of the tumor ALT base, MPQ-filtered VAF, low-MAPQ read ratio
    altcol = np.array([SC[chr(a)] for a in V.alt0[idx]])
    altc = c[np.arange(idx.size), altcol].astype(np.float64)
    mpq_alt = c[np.arange(idx.size), altcol + (SC["MPQ_A"] - SC["A"])].astype(np.float64)
    depth = c[:, SC["DEPTH"]].astype(np.float64); mdepth = c[:, SC["MPQ_DEPTH"]].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        vaf = np.where(depth > 0, altc / depth, 0.0); mvaf = np.where(mdepth > 0, mpq_alt / mdepth, 0.0)
        low = np.where(depth > 0, (depth - mdepth) / depth, 0.0)
    assert np.allclose(vaf, L("norVAF"), rtol=2e-5, atol=1e-7), what + ": normal VAF"
    assert np.allclose(mvaf, L("norMpqVAF"), rtol=2e-5, atol=1e-7), what + ": normal MPQ VAF"
    assert np.allclose(low, L("norMpqReadRatio"), rtol=2e-5, atol=1e-7), what + ": low-MAPQ read ratio"


@pytest.mark.parametrize("name", sorted(fixtures.SOMATIC_FIXTURES))
def test_oracle_normal_extraction_matches_reference_log(name):
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    N, R = util.make_normal_reads(name)
    assert fixtures.input_digest(N) == util.INDEX["somatic:" + name]["normal_digest"]
    V, _, _, _ = util.load_golden_somatic(name)
    out = lps_oracle.somatic_extract_normal(abi.default_params(**over), V, N.ref, R)
    assert V.log_pos.size >= 60
    check_normal_sites(V, out.counters, name)
