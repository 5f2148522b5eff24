"""CPU tests: the oracle restatement (oracle/lps_oracle.cpp) must reproduce, bit for bit, what the REAL reference
binary produced (golden vectors made by tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi


@pytest.mark.parametrize("name", sorted(fixtures.PHASE_FIXTURES))
def test_oracle_matches_reference_golden(name):
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s, V, R = util.make_case(kw)
    assert fixtures.input_digest(s) == util.INDEX[name]["digest"], "generator drift: golden inputs differ"
    gpos, gps, ggt = util.load_golden_phase(name)
    assert np.array_equal(gpos, V.pos)
    out, d = lps_oracle.phase(abi.default_params(**over), V, s.ref, R, dump=True)
    assert d.c.ub_hazard == 0, "fixture touches behaviour that is UB in the reference"
    util.assert_phase_equal(out.phase_set, out.gt, gps, ggt, name)
    assert (out.phase_set != 0).sum() == util.INDEX[name]["n_phased"]


@pytest.mark.parametrize("name", sorted(fixtures.DATA_FIXTURES))
def test_oracle_on_committed_input_files(name):
    """Same check from committed data files (FASTA/VCF/SAM) instead of the generator."""
    kw, cli, over = fixtures.DATA_FIXTURES[name]
    d = os.path.join(util.GOLDEN, "data")
    V = util.parse_vcf_variants(os.path.join(d, f"{name}.vcf"), indels="--indels" in cli)
    R, _ = util.parse_sam(os.path.join(d, f"{name}.sam.gz"))
    ref = util.parse_fasta(os.path.join(d, f"{name}.fa"))
    out, _ = lps_oracle.phase(abi.default_params(**over), V, ref, R)
    import make_golden
    gps, ggt = make_golden.parse_phased_vcf(os.path.join(d, f"{name}.ref_phased.vcf"), V.pos)
    util.assert_phase_equal(out.phase_set, out.gt, gps, ggt, name)
