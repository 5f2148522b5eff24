"""CIGAR shapes around the stream walk of k_extract_phase / k_haplotag_stream (csrc/lps_extract.hip, lps_haplotag.hip): the walk takes the words of
four alignments as one stream of lane-chunks, looks for clips only in the first two and last two words of an alignment and sends everything else
it cannot take - a clip in the middle of a CIGAR, an op of 2^24 bases and more - to the general walker.  Every shape here is compared stage by stage
with the oracle (clip events, observations, edges, votes, result), i.e. with the reference's walk over every op (ParsingBam.cpp:1303-1645)."""
import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu

M, I, D, N, S, H, P, EQ, X = range(9)


def op(code, length):
    return np.uint32((int(length) << 4) | int(code))


def with_cigars(R, new):
    """R with the CIGARs of the reads in `new` ({read: [words]}) replaced; everything else (bases, qualities, starts) stays."""
    arrays = {n: getattr(R, n).copy() for n, _ in abi.Reads.FIELDS}
    per = [R.cigar[int(R.cigar_off[r]):int(R.cigar_off[r + 1])] for r in range(R.n_reads)]
    for r, w in new.items():
        per[r] = np.asarray(w, dtype=np.uint32)
    off = np.zeros(R.n_reads + 1, dtype=np.uint64)
    np.cumsum([len(w) for w in per], out=off[1:])
    arrays["cigar"] = np.concatenate(per).astype(np.uint32) if per else np.zeros(0, np.uint32)
    arrays["cigar_off"] = off
    return abi.Reads(**arrays)


def query_len(words):
    return sum(int(w) >> 4 for w in words if (int(w) & 15) in (M, I, S, EQ, X))


def shapes_for(R):
    """read -> new CIGAR.  Query consumption always equals l_qseq (the reads stay valid records)."""
    new = {}
    lq = R.l_qseq
    live = [r for r in range(R.n_reads) if R.mapq[r] >= 1 and lq[r] > 400]
    assert len(live) > 40
    pick = iter(live[3::3])
    r = next(pick); new[r] = [op(M, lq[r])]                                                                # one word
    r = next(pick); new[r] = [op(S, 6), op(M, lq[r] - 6)]                                                  # two words, a front clip
    r = next(pick); new[r] = [op(M, lq[r] - 9), op(S, 9)]                                                  # two words, a back clip
    r = next(pick); new[r] = [op(S, 7), op(M, lq[r] - 15), op(S, 8)]                                       # three words: both ends, the middle word is neither
    r = next(pick); new[r] = [op(S, 5), op(M, lq[r] - 10), op(S, 5)]                                       # clips of exactly 5: no events, but clip ops
    r = next(pick); new[r] = [op(H, 30), op(S, 6), op(M, lq[r] - 13), op(S, 7), op(H, 12)]                  # H S ... S H
    r = next(pick); new[r] = [op(H, 3), op(M, lq[r]), op(H, 40)]                                           # hard clips only, one short one long
    r = next(pick); new[r] = [op(S, 10), op(M, 100), op(D, 3), op(M, lq[r] - 130), op(I, 2), op(M, 10), op(S, 8)]   # seven words: ends + interior
    r = next(pick); new[r] = [op(S, 6), op(S, 7), op(M, lq[r] - 13)]                                       # two clips in a row at the front (index 0 and 1)
    r = next(pick); new[r] = [op(M, lq[r] - 20), op(S, 12), op(S, 8)]                                      # ... and at the back
    r = next(pick); new[r] = [op(M, 200), op(S, 9), op(M, lq[r] - 209)]                                    # a clip in the MIDDLE (3 words: index 1 is an end word)
    r = next(pick); new[r] = [op(M, 100), op(I, 1), op(M, 100), op(S, 11), op(M, lq[r] - 212)]             # a clip in the middle of five words: general walker
    r = next(pick); new[r] = [op(M, 50), op(H, 9), op(M, 50), op(D, 2), op(M, lq[r] - 100)]                # a hard clip in the middle
    r = next(pick); new[r] = [op(M, 150), op(N, (1 << 24) + 3), op(M, lq[r] - 150)]                        # one op of 2^24 bases and more (general walker)
    r = next(pick); new[r] = [op(EQ, 100), op(X, 1), op(EQ, lq[r] - 101)]                                  # = and X
    r = next(pick); new[r] = [op(M, 60), op(P, 4), op(M, lq[r] - 60)]                                      # a padding op
    for r, w in new.items():
        assert query_len(w) == int(lq[r]), (r, w)
    return new


@pytest.mark.parametrize("name", ["snp_ont", "two_blocks"])
def test_phase_with_crafted_cigars_matches_the_oracle_stage_by_stage(name):
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s, V, R0 = util.make_case(kw)
    P = abi.default_params(**over)
    new = shapes_for(R0)
    R = with_cigars(R0, new)
    ref_out, d = lps_oracle.phase(P, V, s.ref, R, dump=True)
    with hip.Context(0, P) as ctx:
        out = ctx.phase(V, s.ref, R)
        util.assert_stages_equal(ctx, d, name + " with crafted CIGARs")
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, name + " with crafted CIGARs")
        # the same alignments with a filtered one (MAPQ 0) that holds a clip in the middle between walked ones: its words pass by in the stream
        arrays = {n: getattr(R, n).copy() for n, _ in abi.Reads.FIELDS}
        mid = [r for r, w in new.items() if len(w) == 5 and (int(w[3]) & 15) == S][0]
        arrays["mapq"][mid] = 0
        R2 = abi.Reads(**arrays)
        ref2, d2 = lps_oracle.phase(P, V, s.ref, R2, dump=True)
        out2 = ctx.phase(V, s.ref, R2)
        util.assert_stages_equal(ctx, d2, name + " with a filtered alignment that holds an interior clip")
        util.assert_phase_equal(out2.phase_set, out2.gt, ref2.phase_set, ref2.gt, name + " filtered interior clip")
    assert d.c.n_clips > 0


def test_haplotag_with_crafted_cigars_matches_the_oracle():
    kw, cli, over = fixtures.PHASE_FIXTURES["two_blocks"]
    s, V, R0 = util.make_case(kw)
    P = abi.default_params()
    new = shapes_for(R0)
    big = [r for r, w in new.items() if any((int(x) >> 4) >= (1 << 24) for x in w)]
    assert len(big) == 1
    with hip.Context(0, P) as ctx:
        ph = ctx.phase(V, s.ref, R0)
        idx = np.nonzero(ph.phase_set != 0)[0]
        VT = abi.Variants(V.pos[idx], [V.ref_str[i] for i in idx], [V.alt_str[i] for i in idx], hp1_is_alt=ph.gt[idx], phase_set=ph.phase_set[idx])
        # one op of 2^24 bases and more is outside the stream walk's 24-bit sums: the chromosome is then scored by the per-op-prefix walker ...
        Rb = with_cigars(R0, new)
        outb = ctx.haplotag(VT, s.ref, Rb)
        # ... everything else by the stream walk
        ok = {r: w for r, w in new.items() if r not in big}
        R = with_cigars(R0, ok)
        out = ctx.haplotag(VT, s.ref, R)
    for o, rr in ((outb, Rb), (out, R)):
        ref = lps_oracle.haplotag(P, VT, s.ref, rr)
        for k in ("status", "hp1", "hp2", "ps_min", "hp", "pq", "ps"):
            assert np.array_equal(getattr(o, k), getattr(ref, k)), k
    assert (out.hp != 0).sum() > 0.4 * R.n_reads
