"""GPU parity tests of the somatic_haplotag tagging pass (row a22) through the C-ABI: integer counts == CPU oracle,
HP:Z / PS / PQ == the tags the reference binary wrote to the tagged tumor BAM."""
import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(fixtures.SOMATIC_FIXTURES))
def test_somatic_tag_matches_oracle_and_reference(name):
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    T, R = util.make_tumor_reads(name)
    V, hp, ps, pq = util.load_golden_somatic(name)
    P = abi.default_params(**over)
    ref = lps_oracle.somatic_tag(P, V, R)
    with hip.Context(0, P) as ctx:
        out = ctx.somatic_tag(V, T.ref, R)
    for k in ("status", "hp1", "hp2", "hp3", "derive_h1", "derive_h2", "ps_min", "hp", "pq", "ps"):
        assert np.array_equal(getattr(out, k), getattr(ref, k)), k
    assert np.array_equal(np.minimum(out.n_ps, 2), np.minimum(ref.n_ps, 2))
    util.assert_somatic_tags_equal(out, hp, ps, pq, name + " vs reference BAM tags")
    assert (out.hp >= 5).sum() > 0, "fixture should contain somatic (H1-1 / H2-1) reads"


@pytest.mark.parametrize("name", sorted(fixtures.SOMATIC_FIXTURES))
def test_normal_extraction_matches_oracle_and_reference_log(name):
    """Row a20: per-site counters of the normal-BAM pass == oracle for EVERY table row, and == the reference's log at its calls."""
    from test_oracle_somatic_extract_golden import check_normal_sites
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    N, R = util.make_normal_reads(name)
    V, _, _, _ = util.load_golden_somatic(name)
    P = abi.default_params(**over)
    ref = lps_oracle.somatic_extract_normal(P, V, N.ref, R)
    with hip.Context(0, P) as ctx:
        out = ctx.somatic_extract_normal(V, N.ref, R)
    assert np.array_equal(out.read_hp, ref.read_hp), "per-read germline haplotype of the pass differs"
    bad = np.nonzero((out.counters != ref.counters).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} sites differ, first {bad[:5]}: {out.counters[bad[:3]]} vs {ref.counters[bad[:3]]}"
    check_normal_sites(V, out.counters, name + " vs reference log")


@pytest.mark.parametrize("name", sorted(fixtures.SOMATIC_FIXTURES))
def test_tumor_extraction_matches_oracle_and_reference_logs(name):
    """Row a21: every per-site counter, per-read record, (site, read, base HP) pair and +-100 bp difference-window entry == oracle
    (lists compared as sorted multisets: their order is not defined), and the reference's logged fields / DenseAlt counts."""
    from test_oracle_somatic_tumor_golden import check_tumor_sites, dense_alt_same_count
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    T, R = util.make_tumor_reads(name)
    V, _, _, _ = util.load_golden_somatic(name)
    P = abi.default_params(**over)
    ref = lps_oracle.somatic_extract_tumor(P, V, T.ref, R)
    with hip.Context(0, P) as ctx:
        out = ctx.somatic_extract_tumor(V, T.ref, R, pair_cap=16, win_cap=16)     # forces the capacity-retry path once
    for k in ("status", "hp1", "hp2", "hp3", "hp", "ps_min", "end_pos", "read_len", "has_site"):
        assert np.array_equal(getattr(out, k), getattr(ref, k)), k
    assert np.array_equal(np.minimum(out.n_ps, 2), np.minimum(ref.n_ps, 2))
    bad = np.nonzero((out.site != ref.site).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} sites differ, first {bad[:3]}: {out.site[bad[:2]]} vs {ref.site[bad[:2]]}"
    assert out.c.n_pairs == ref.c.n_pairs and out.c.n_windows == ref.c.n_windows
    for a, b in zip(out.pairs(), ref.pairs()):
        assert np.array_equal(a, b), "pairs differ"
    for a, b in zip(out.windows(), ref.windows()):
        assert np.array_equal(a, b), "difference windows differ"
    check_tumor_sites(V, out, name + " vs reference log")
    same = dense_alt_same_count(V, out)
    didx = np.searchsorted(V.pos, V.dense_pos)
    assert np.array_equal(same[didx], V.dense_cnt), "DenseAlt sameCount vs reference log"


def test_an_operation_of_2_pow_24_bases_sends_every_somatic_pass_to_the_general_walker():
    """The three passes run on the stream walker (k_haplotag_stream<1>, <2,3>, k_tumor_stream); one CIGAR operation of 2^24 bases and more is outside
    its 24-bit sums: the chromosome is then taken again by the per-op-prefix walkers (k_haplotag_score<MODE>, k_tumor_extract<PASS>), same results."""
    from test_cigar_shapes_gpu import with_cigars, op, M, N
    name = sorted(fixtures.SOMATIC_FIXTURES)[0]
    genome, nkw, tkw, cli, over = fixtures.SOMATIC_FIXTURES[name]
    V, _, _, _ = util.load_golden_somatic(name)
    P = abi.default_params(**over)
    T, RT0 = util.make_tumor_reads(name)
    Nn, RN0 = util.make_normal_reads(name)

    def crafted(R):
        lq = np.asarray(R.l_qseq)
        r = int(np.nonzero(lq > 400)[0][R.n_reads // 3 % max(1, int((lq > 400).sum()))])
        return with_cigars(R, {r: [op(M, 150), op(N, (1 << 24) + 3), op(M, int(lq[r]) - 150)]})
    RT, RN = crafted(RT0), crafted(RN0)
    with hip.Context(0, P) as ctx:
        o1 = ctx.somatic_extract_normal(V, Nn.ref, RN)
        o2 = ctx.somatic_extract_tumor(V, T.ref, RT)
        o3 = ctx.somatic_tag(V, T.ref, RT)
    w1 = lps_oracle.somatic_extract_normal(P, V, Nn.ref, RN)
    assert np.array_equal(o1.read_hp, w1.read_hp) and np.array_equal(o1.counters, w1.counters)
    w2 = lps_oracle.somatic_extract_tumor(P, V, T.ref, RT)
    for k in ("status", "hp1", "hp2", "hp3", "hp", "ps_min", "end_pos", "read_len", "has_site"):
        assert np.array_equal(getattr(o2, k), getattr(w2, k)), k
    assert np.array_equal(o2.site, w2.site) and o2.c.n_pairs == w2.c.n_pairs and o2.c.n_windows == w2.c.n_windows
    assert all(np.array_equal(a, b) for a, b in zip(o2.pairs(), w2.pairs())) and all(np.array_equal(a, b) for a, b in zip(o2.windows(), w2.windows()))
    w3 = lps_oracle.somatic_tag(P, V, RT)
    for k in ("status", "hp1", "hp2", "hp3", "derive_h1", "derive_h2", "ps_min", "hp", "pq", "ps"):
        assert np.array_equal(getattr(o3, k), getattr(w3, k)), k
