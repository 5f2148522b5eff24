"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the reference's golden vectors.
Bit-exact for genotypes, phase sets, observations, nodes, votes; the fp32 edge matrix is compared exactly too
(the kernels replay the reference's accumulation order)."""
import numpy as np
import pytest

import fixtures
import lps_oracle
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu


def run_both(kw, over):
    s, V, R = util.make_case(kw)
    P = abi.default_params(**over)
    ref_out, d = lps_oracle.phase(P, V, s.ref, R, dump=True)
    ctx = hip.Context(0, P)
    out = ctx.phase(V, s.ref, R)
    return s, V, R, P, ref_out, d, ctx, out


@pytest.mark.parametrize("name", sorted(fixtures.PHASE_FIXTURES))
def test_phase_matches_oracle_and_golden(name):
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s, V, R, P, ref_out, d, ctx, out = run_both(kw, over)
    try:
        # stage: observations (per alignment, in position order)
        cnt, var, al, q = ctx.dump_observations()
        n = d.c.n_obs
        assert np.array_equal(cnt, d.obs_count), "per-read observation counts differ"
        assert np.array_equal(var, d.obs_var[:n]) and np.array_equal(al, d.obs_allele[:n])
        assert np.array_equal(q.astype(np.int32), d.obs_quality[:n].astype(np.int32))
        # stage: clips + CNV intervals + overlap filter
        cp, cf = ctx.dump_clips()
        o = np.lexsort((d.clip_fb[:d.c.n_clips], d.clip_pos[:d.c.n_clips]))
        assert np.array_equal(cp, d.clip_pos[:d.c.n_clips][o]) and np.array_equal(cf, d.clip_fb[:d.c.n_clips][o])
        cs, ce, dele = ctx.dump_cnv()
        assert list(cs) == list(d.cnv_start()) and list(ce) == list(d.cnv_end())
        assert np.array_equal(dele, d.aln_deleted)
        # stage: graph nodes + fp32 edge matrix (exact: same accumulation order)
        nodes, edge = ctx.dump_graph()
        N = d.c.n_nodes
        assert np.array_equal(nodes, d.node_var[:N])
        assert np.array_equal(edge.view(np.uint32), d.edge[:N].view(np.uint32)), "edge matrix differs bitwise"
        # stage: vote scan
        hp, blk = ctx.dump_votes()
        assert np.array_equal(hp, d.node_hp[:N]) and np.array_equal(blk, d.node_block[:N])
        # final result vs oracle and vs the reference binary's golden output
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, name + " vs oracle")
        gpos, gps, ggt = util.load_golden_phase(name)
        util.assert_phase_equal(out.phase_set, out.gt, gps, ggt, name + " vs reference golden")
    finally:
        ctx.close()


@pytest.mark.parametrize("every", [1, 2, 3, 7, 29])
@pytest.mark.parametrize("name", ["snp_ont_seed2", "cnv_many", "cnv_64"])
def test_vote_scan_serial_replay_path(name, every, monkeypatch):
    """k_scan_stitch's fallback: boundaries declared unmatched (LPS_SCAN_FORCE_REPLAY) are replayed serially from the true state -
    the votes and the result must not change."""
    kw, cli, over = fixtures.PHASE_FIXTURES[name]
    s, V, R, P, ref_out, d, ctx, out = run_both(kw, over)
    try:
        hp0, blk0 = ctx.dump_votes()
        monkeypatch.setenv("LPS_SCAN_FORCE_REPLAY", str(every))
        out2 = ctx.phase(V, s.ref, R)
        assert ctx.timings()["n_scan_replayed"] > 0
        hp1, blk1 = ctx.dump_votes()
        N = d.c.n_nodes
        assert np.array_equal(hp1, d.node_hp[:N]) and np.array_equal(blk1, d.node_block[:N])
        assert np.array_equal(hp0, hp1) and np.array_equal(blk0, blk1)
        util.assert_phase_equal(out2.phase_set, out2.gt, ref_out.phase_set, ref_out.gt, name + " replayed vs oracle")
    finally:
        ctx.close()


@pytest.mark.parametrize("seed,len_median", [(71, 15000.0), (72, 3000.0)])
def test_rows_longer_than_the_extraction_buffer(seed, len_median):
    """k_extract_phase collects a wave's observations (four alignments) in a 512-entry LDS buffer; a SNP every ~10 bp makes single rows far
    longer than that, so the buffer-full path (reserve an upper bound, flush, write the rest of the row directly) runs, also twice per wave."""
    kw = dict(seed=seed, contig_len=150_000, n_snp=15_000, coverage=12.0, len_median=len_median, len_min=500, sub_rate=0.03,
              lowq_frac=0.1, supp_frac=0.1, n_threads=2, snp_pair_frac=0.05, snp_in_hpoly_frac=0.1, hpoly_every=400.0)
    s, V, R, P, ref_out, d, ctx, out = run_both(kw, {})
    try:
        cnt, var, al, q = ctx.dump_observations()
        assert cnt.max() > 512
        n = d.c.n_obs
        assert np.array_equal(cnt, d.obs_count), "per-read observation counts differ"
        assert np.array_equal(var, d.obs_var[:n]) and np.array_equal(al, d.obs_allele[:n])
        assert np.array_equal(q.astype(np.int32), d.obs_quality[:n].astype(np.int32))
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, "dense rows vs oracle")
    finally:
        ctx.close()


def test_nodes_with_more_than_64_reads():
    """Coverage ~90: every node's entry list is longer than a wavefront, so k_edges orders it through memory (rank sort) instead of in registers;
    the fp32 edge sums must still be the reference's bit for bit."""
    kw = dict(seed=81, contig_len=80_000, n_snp=90, coverage=90.0, len_median=9000.0, len_min=1000, sub_rate=0.03, lowq_frac=0.1, supp_frac=0.05,
              n_threads=2, snp_pair_frac=0.02, snp_in_hpoly_frac=0.1, hpoly_every=400.0)
    s, V, R, P, ref_out, d, ctx, out = run_both(kw, {})
    try:
        nodes, edge = ctx.dump_graph()
        N = d.c.n_nodes
        assert np.array_equal(nodes, d.node_var[:N])
        assert np.array_equal(edge.view(np.uint32), d.edge[:N].view(np.uint32)), "edge matrix differs bitwise"
        cnt, var, al, q = ctx.dump_observations()
        assert np.bincount(var).max() > 64
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, "coverage 90 vs oracle")
    finally:
        ctx.close()


def test_table_of_more_than_2p22_rows():
    """A variant table beyond the 4 194 303 rows the first versions could address (the hit word kept the row in 22 bits; the reference's std::map
    has no limit, src/phase/ParsingBam.h:165-185): 4.3 M more rows behind the region the reads cover.  Result == oracle."""
    import lps_oracle
    from lps import abi, hip
    from lps.synth import Synth
    s = Synth(seed=83, contig_len=300_000, n_snp=350, coverage=14.0, n_threads=2)
    extra = 4_300_000
    pos = np.concatenate([np.asarray(s.var_pos, np.int32), (300_010 + 3 * np.arange(extra)).astype(np.int32)])
    ref = np.concatenate([np.asarray(s.ref, np.uint8), np.full(3 * extra + 64, ord("A"), np.uint8)])
    V = abi.Variants(pos, list(s.var_ref) + [b"A"] * extra, list(s.var_alt) + [b"C"] * extra)
    assert V.n > (1 << 22)
    R = abi.Reads.from_synth(s)
    P = abi.default_params()
    with hip.Context(0, P) as ctx:
        out = ctx.phase(V, ref, R)
    want, _ = lps_oracle.phase(P, V, ref, R)
    util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "4.3 M-row table vs oracle")
    assert int((out.phase_set != 0).sum()) > 300


def test_observations_of_one_variant_beyond_the_old_16_bit_rank():
    """More than 65 536 alignments over one variant (ultra-deep amplicon data): the rank of an observation inside its variant's list was a 16-bit
    field once (k_graph_obs raised LPS_ERR_KEY_RANGE); it has 22 bits now and the limit has its own message.  84 000 short reads over 7 SNPs."""
    import lps_oracle
    from lps import abi, hip
    from lps.synth import Synth
    s = Synth(seed=84, contig_len=3_000, n_snp=6, coverage=75000.0, len_median=2000.0, len_min=1800, len_max=2600, n_threads=4, clip_every=7)
    assert s.n_reads > 80_000
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt); R = abi.Reads.from_synth(s)
    P = abi.default_params()
    with hip.Context(0, P) as ctx:
        out = ctx.phase(V, s.ref, R)
        cnt, var, al, q = ctx.dump_observations()
        assert np.bincount(var).max() > 65_536
    want, _ = lps_oracle.phase(P, V, s.ref, R)
    util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "81 000 observations of one variant vs oracle")


def test_unsupported_cigar_op_is_an_error():
    """The reference prints "alignment find unsupported CIGAR operation" and exits (ParsingBam.cpp:1625-1628); the library returns an error
    with that message.  An op code above 8 in an alignment that is filtered out (MAPQ 0) is never looked at, as in the reference."""
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont_seed2"]
    s, V, R = util.make_case(kw)
    P = abi.default_params(**over)
    r = R.n_reads // 2
    o = int(R.cigar_off[r]) + (int(R.cigar_off[r + 1]) - int(R.cigar_off[r])) // 2
    arrays = {n: getattr(R, n).copy() for n, _ in abi.Reads.FIELDS}
    arrays["cigar"][o] = (arrays["cigar"][o] & ~np.uint32(15)) | np.uint32(9)          # op code 9 ('B')
    bad = abi.Reads(**arrays)
    ctx = hip.Context(0, P)
    try:
        with pytest.raises(hip.LpsError, match="unsupported CIGAR"):
            ctx.phase(V, s.ref, bad)
        arrays["mapq"][r] = 0                                                          # filtered out: its CIGAR is not walked
        skipped = abi.Reads(**arrays)
        out = ctx.phase(V, s.ref, skipped)
        ref_out, _ = lps_oracle.phase(P, V, s.ref, R.subset(np.delete(np.arange(R.n_reads), r)))
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, "bad op in a filtered alignment")
    finally:
        ctx.close()


def test_repeat_runs_are_identical_and_recomputed():
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont_seed2"]
    s, V, R = util.make_case(kw)
    with hip.Context(0, abi.default_params()) as ctx:
        ctx.load_chromosome(V, s.ref, R)
        a = ctx.run_phase()
        b = ctx.run_phase()
        assert np.array_equal(a.phase_set, b.phase_set) and np.array_equal(a.gt, b.gt)
        t = ctx.timings()
        assert t["n_obs"] > 0 and t["stages"]["extract"] > 0


def test_batched_push_equals_single_push():
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont"]
    s, V, R = util.make_case(kw)
    n = R.n_reads
    parts = [R.subset(np.arange(0, n // 3)), R.subset(np.arange(n // 3, n // 2)), R.subset(np.arange(n // 2, n))]
    with hip.Context(0, abi.default_params()) as ctx:
        a = ctx.phase(V, s.ref, R)
        b = ctx.phase(V, s.ref, parts)
        assert np.array_equal(a.phase_set, b.phase_set) and np.array_equal(a.gt, b.gt)


def test_batches_pushed_out_of_order_are_refused():
    """Every kernel assumes coordinate order over ALL resident alignments: a batch that starts before the end of the one pushed before it is an
    error like an unsorted batch (the order inside a batch is checked by lps_push_reads / k_batch_check)."""
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont"]
    s, V, R = util.make_case(kw)
    n = R.n_reads
    assert R.ref_start[n // 2] > R.ref_start[0]
    parts = [R.subset(np.arange(n // 2, n)), R.subset(np.arange(0, n // 2))]
    with hip.Context(0, abi.default_params()) as ctx:
        with pytest.raises(hip.LpsError, match="coordinate-sorted"):
            ctx.phase(V, s.ref, parts)
        a = ctx.phase(V, s.ref, R)                  # the context is usable afterwards
        assert (a.phase_set != 0).any()


def test_empty_and_degenerate_inputs():
    kw, cli, over = fixtures.PHASE_FIXTURES["snp_ont"]
    s, V, R = util.make_case(kw)
    with hip.Context(0, abi.default_params()) as ctx:
        # no reads at all
        out = ctx.phase(V, s.ref, R.subset(np.zeros(0, np.int64)))
        assert (out.phase_set == 0).all()
        # all reads filtered by MAPQ
        ctx2 = hip.Context(0, abi.default_params(mapping_quality=61))
        out = ctx2.phase(V, s.ref, R)
        assert (out.phase_set == 0).all()
        ctx2.close()
        # a single read cannot phase anything but must not crash
        one = R.subset(np.array([np.argmax(R.l_qseq)]))
        ref_out, _ = lps_oracle.phase(abi.default_params(), V, s.ref, one)
        out = ctx.phase(V, s.ref, one)
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, "single read")


def test_config1_scale_matches_oracle():
    """BASELINE.json configs[0]: 5 Mb, 10x, ~5k het SNPs."""
    s, V, R = util.make_case(dict(seed=1, contig_len=5_000_000, n_snp=5000, coverage=10.0))
    P = abi.default_params()
    ref_out, _ = lps_oracle.phase(P, V, s.ref, R)
    with hip.Context(0, P) as ctx:
        out = ctx.phase(V, s.ref, R)
    util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, "config1")
    assert (out.phase_set != 0).sum() > 4900


def test_clip_sort_path_on_inputs_that_skip_it():
    """Since round 4 the clip keys are only sorted (and the CNV state machine only replayed) when some (position, end) holds five clips - the
    count-min bound k_name_link takes.  LPS_CLIP_ALWAYS_SORT=1 forces the sorted path on a fixture that skips it: same stages, same result.
    (The variable is read once per process: a child process.)"""
    import os, subprocess, sys
    code = ("import sys; sys.path[:0] = [%r, %r, %r, %r]\n"
            "import numpy as np, fixtures, util, lps_oracle\nfrom lps import abi, hip\n"
            "kw, _, over = fixtures.PHASE_FIXTURES['snp_ont']\ns, V, R = util.make_case(kw)\nP = abi.default_params(**over)\n"
            "want, d = lps_oracle.phase(P, V, s.ref, R, dump=True)\n"
            "with hip.Context(0, P) as ctx:\n    out = ctx.phase(V, s.ref, R)\n    util.assert_stages_equal(ctx, d, 'always-sort')\n"
            "util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, 'always-sort')\nprint('ok')\n") % (
        os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"),
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "longphase-s_amd"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LPS_CLIP_ALWAYS_SORT="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]
