"""CPU tests: the somatic-tagging restatement reproduces the HP:Z / PS / PQ tags the REAL reference binary wrote to the
tagged tumor BAM (tests/golden/somatic_tag_*.npz: reference phase on the normal sample -> reference somatic_haplotag;
the isSomaticVariant / somaticReadDeriveByHP inputs are taken from the reference's own _sc.vcf and read-distribution log)."""
import pytest

import fixtures
import lps_oracle
import util
from lps import abi


@pytest.mark.parametrize("name", sorted(fixtures.SOMATIC_FIXTURES))
def test_oracle_somatic_tag_matches_reference_tags(name):
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    T, R = util.make_tumor_reads(name)
    assert fixtures.input_digest(T) == util.INDEX["somatic:" + name]["digest"]
    V, hp, ps, pq = util.load_golden_somatic(name)
    out = lps_oracle.somatic_tag(abi.default_params(**over), V, R)
    util.assert_somatic_tags_equal(out, hp, ps, pq, name)
