"""lps_push_bam_records (raw BAM records decoded on the GPU) against lps_push_reads (host-decoded SoA)
and the reference's golden output: observations, phase result and haplotag tags must be identical, whatever the byte
alignment of the CIGAR/seq/qual fields inside the records."""
import numpy as np
import pytest

import fixtures
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["snp_ont", "indels", "supp_overlap", "cnv_pileup"])
def test_phase_from_bam_records(name):
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s, V, R = util.make_case(kw)
    P = abi.default_params(**over)
    B = abi.BamRecords.from_reads(R, seed=7, lead=5)
    ctx = hip.Context(0, P)
    try:
        a = ctx.phase(V, s.ref, R)
        obs_a = ctx.dump_observations()
        b = ctx.phase(V, s.ref, B)
        obs_b = ctx.dump_observations()
        for x, y in zip(obs_a, obs_b):
            assert np.array_equal(x, y)
        util.assert_phase_equal(b.phase_set, b.gt, a.phase_set, a.gt, name + " BAM records vs SoA")
        gpos, gps, ggt = util.load_golden_phase(name)
        util.assert_phase_equal(b.phase_set, b.gt, gps, ggt, name + " BAM records vs reference golden")
    finally:
        ctx.close()


def test_two_pushes_equal_one():
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont_seed2"]
    s, V, R = util.make_case(kw)
    B = abi.BamRecords.from_reads(R, seed=3)
    k = R.n_reads // 3
    cut = int(B.rec_off[k]) - 4
    B1 = abi.BamRecords(B.blob[:cut], B.rec_off[:k], B.name_id[:k])
    B2 = abi.BamRecords(B.blob[cut:], B.rec_off[k:] - np.uint64(cut), B.name_id[k:])
    ctx = hip.Context(0, abi.default_params(**over))
    try:
        one = ctx.phase(V, s.ref, B)
        two = ctx.phase(V, s.ref, [B1, B2])
        assert np.array_equal(one.phase_set, two.phase_set) and np.array_equal(one.gt, two.gt)
        gpos, gps, ggt = util.load_golden_phase("snp_ont_seed2")
        util.assert_phase_equal(two.phase_set, two.gt, gps, ggt, "two pushes vs golden")
    finally:
        ctx.close()


def test_haplotag_from_bam_records():
    name = "snp_ont"
    src, cli, over = fixtures.HAPLOTAG_FIXTURES[name]
    s, _, R = util.make_case(fixtures.PHASE_FIXTURES[src][0])
    V, hp, ps, pq = util.load_golden_haplotag(name)
    ctx = hip.Context(0, abi.default_params(**over))
    try:
        out = ctx.haplotag(V, s.ref, abi.BamRecords.from_reads(R, seed=11, lead=3))
        util.assert_tags_equal(out, hp, ps, pq, name)
    finally:
        ctx.close()


def test_bad_records_are_rejected():
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont"]
    s, V, R = util.make_case(kw)
    B = abi.BamRecords.from_reads(R, seed=1)
    ctx = hip.Context(0, abi.default_params())
    try:
        trunc = abi.BamRecords(B.blob[: int(B.rec_off[-1]) + 40], B.rec_off, B.name_id)          # last record cut short
        with pytest.raises(hip.LpsError, match="does not fit"):
            ctx.phase(V, s.ref, trunc)
        off = B.rec_off.copy(); off[[10, 11]] = off[[11, 10]]
        with pytest.raises(hip.LpsError, match="does not fit|coordinate-sorted"):
            ctx.phase(V, s.ref, abi.BamRecords(B.blob, off, B.name_id))
        with pytest.raises(hip.LpsError, match="same chromosome"):
            ctx.phase(V, s.ref, [R, B])
        out = ctx.phase(V, s.ref, B)                                                              # the ctx is still usable
        gpos, gps, ggt = util.load_golden_phase("snp_ont")
        util.assert_phase_equal(out.phase_set, out.gt, gps, ggt, "after rejected pushes")
    finally:
        ctx.close()
