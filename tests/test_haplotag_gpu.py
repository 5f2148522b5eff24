"""GPU parity tests of the haplotag scorer (through the C-ABI) against the CPU oracle and the tags the reference
binary wrote (golden).  Integer counts, PS and PQ must be identical."""
import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(fixtures.HAPLOTAG_FIXTURES))
def test_haplotag_matches_oracle_and_reference_tags(name):
    src, cli, over = fixtures.HAPLOTAG_FIXTURES[name]
    kw, _, _ = fixtures.PHASE_FIXTURES[src]
    s, _, R = util.make_case(kw)
    V, hp, ps, pq = util.load_golden_haplotag(name)
    P = abi.default_params(**over)
    ref = lps_oracle.haplotag(P, V, s.ref, R)
    with hip.Context(0, P) as ctx:
        out = ctx.haplotag(V, s.ref, R)
    assert np.array_equal(out.status, ref.status)
    assert np.array_equal(out.hp1, ref.hp1) and np.array_equal(out.hp2, ref.hp2), "vote counts differ"
    assert np.array_equal(np.minimum(out.n_ps, 2), np.minimum(ref.n_ps, 2))
    assert np.array_equal(out.ps_min, ref.ps_min)
    assert np.array_equal(out.hp, ref.hp) and np.array_equal(out.pq, ref.pq) and np.array_equal(out.ps, ref.ps)
    util.assert_tags_equal(out, hp, ps, pq, name + " vs reference BAM tags")


@pytest.mark.parametrize("name", ["snp_ont", "sparse_cov"])
def test_haplotag_with_read_votes(name):
    """lps_set_read_votes (judgeSVHap): votes from the phased SV / MOD files enter every scored alignment's counts before the decision - also for
    alignments without any SNP vote (they are tagged with PS 0, as the reference does); votes of a previous contig do not leak."""
    src, cli, over = fixtures.HAPLOTAG_FIXTURES[name]
    kw, _, _ = fixtures.PHASE_FIXTURES[src]
    s, _, R = util.make_case(kw)
    V, _, _, _ = util.load_golden_haplotag(name)
    P = abi.default_params(**over)
    g = np.random.default_rng(3)
    v1 = (g.integers(0, 30, R.n_reads) * (g.random(R.n_reads) < 0.4)).astype(np.int32)
    v2 = (g.integers(0, 30, R.n_reads) * (g.random(R.n_reads) < 0.4)).astype(np.int32)
    plain = lps_oracle.haplotag(P, V, s.ref, R)
    lone = np.nonzero((plain.status == 0) & (plain.n_ps == 0))[0]        # scored alignments that saw no phased SNP: votes alone tag them
    v1[lone] = 3; v2[lone] = 0
    want = lps_oracle.haplotag(P, V, s.ref, R, votes=(v1, v2))
    assert (want.hp != plain.hp).sum() >= 5 and (want.pq != plain.pq).sum() >= 20
    assert np.all(want.hp[lone] == 1) and np.all(want.ps[lone] == 0)
    with hip.Context(0, P) as ctx:
        ctx.load_chromosome(V, s.ref, R)
        ctx.set_read_votes(v1, v2)
        out = ctx.run_haplotag()
        for k in ("status", "hp1", "hp2", "ps_min", "hp", "pq", "ps"):
            assert np.array_equal(getattr(out, k), getattr(want, k)), k
        out2 = ctx.haplotag(V, s.ref, R)                       # a new chromosome: no votes
        for k in ("status", "hp1", "hp2", "hp", "pq", "ps"):
            assert np.array_equal(getattr(out2, k), getattr(plain, k)), k
        assert ctx.L.lps_set_read_votes(ctx.h, v1[:5].ctypes.data, v2[:5].ctypes.data, 5) != 0


def test_haplotag_after_own_phase_roundtrip():
    """phase on the GPU, build the phased table from its result, haplotag on the GPU == oracle on the same table."""
    kw, cli, over = fixtures.PHASE_FIXTURES["two_blocks"]
    s, V, R = util.make_case(kw)
    P = abi.default_params()
    with hip.Context(0, P) as ctx:
        ph = ctx.phase(V, s.ref, R)
        m = ph.phase_set != 0
        idx = np.nonzero(m)[0]
        VT = abi.Variants(V.pos[idx], [V.ref_str[i] for i in idx], [V.alt_str[i] for i in idx],
                          hp1_is_alt=ph.gt[idx], phase_set=ph.phase_set[idx])
        out = ctx.haplotag(VT, s.ref, R)
    ref = lps_oracle.haplotag(P, VT, s.ref, R)
    for k in ("status", "hp1", "hp2", "ps_min", "hp", "pq", "ps"):
        assert np.array_equal(getattr(out, k), getattr(ref, k)), k
    # tagged reads must agree with the simulated molecule haplotype up to one global flip per phase set
    tagged = out.hp != 0
    assert tagged.sum() > 0.5 * R.n_reads
    for psv in np.unique(out.ps[tagged]):
        sel = tagged & (out.ps == psv)
        agree = ((out.hp[sel] - 1) == s.read_hap[sel]).mean()
        assert agree > 0.97 or agree < 0.03, (psv, agree)


def test_haplotag_empty_table_and_filters():
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont"]
    s, V, R = util.make_case(kw)
    empty = abi.Variants(np.zeros(0, np.int32), [], [], hp1_is_alt=np.zeros(0, np.uint8), phase_set=np.zeros(0, np.int32))
    P = abi.default_params()
    ref = lps_oracle.haplotag(P, empty, s.ref, R)
    assert set(np.unique(ref.status)) <= {1, 3, 4, 5}
