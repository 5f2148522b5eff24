"""CPU tests: the haplotag restatement of the oracle reproduces the HP/PS/PQ tags the REAL reference binary wrote
(tests/golden/haplotag_*.npz, made by make_golden.py: reference phase -> reference haplotag -> tagged BAM)."""
import pytest

import fixtures
import lps_oracle
import util
from lps import abi


@pytest.mark.parametrize("name", sorted(fixtures.HAPLOTAG_FIXTURES))
def test_oracle_haplotag_matches_reference_tags(name):
    src, cli, over = fixtures.HAPLOTAG_FIXTURES[name]
    kw, _, _ = fixtures.PHASE_FIXTURES[src]
    s, _, R = util.make_case(kw)
    assert fixtures.input_digest(s) == util.INDEX["haplotag:" + name]["digest"]
    V, hp, ps, pq = util.load_golden_haplotag(name)
    out = lps_oracle.haplotag(abi.default_params(**over), V, s.ref, R)
    util.assert_tags_equal(out, hp, ps, pq, name)
    assert int((out.hp != 0).sum()) == util.INDEX["haplotag:" + name]["n_tagged"]
