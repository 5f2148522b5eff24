"""SV and MOD rows co-phased with the SNPs (`phase --sv-file / --mod-file`; get_snp's SV branch src/phase/ParsingBam.cpp:1397-1434 and MOD
branch :1373-1395, node types 1 / 2 in addEdge, the SNP<->MOD edge threshold of findBestEdgePair, readCorrection's SV / MOD cases): the HIP
path through the C-ABI against the CPU oracle - every stage dump in indices of the union of the three tables, the results of all three.
The oracle walks the reference's three cursors literally and is itself held against the reference binary (tests/test_oracle_extra_golden.py);
the GPU serves the rows by the closed form in csrc/lps_extra.hip."""
import numpy as np
import pytest

import lps_oracle
import util
from lps import abi, hip
from lps.synth import Synth, make_mod_lines, merge_mod_lines

pytestmark = pytest.mark.gpu

BASE = dict(contig_len=400_000, n_snp=500, coverage=15.0, n_threads=4)

CASES = {
    # name: (synth kwargs, mod kwargs or None, use SVs, params, extra kwargs)
    "sv_and_mod": (dict(BASE, seed=21, sv_every=15000.0), dict(), True, {}, {}),
    "sv_only": (dict(BASE, seed=22, sv_every=15000.0), None, True, {}, {}),
    "mod_only": (dict(BASE, seed=23), dict(), False, {}, {}),
    "pb_indels": (dict(BASE, seed=24, sv_every=15000.0, indel_var_frac=0.3), dict(), True, dict(is_ont=0, phase_indel=1), {}),
    "supp_cnv": (dict(BASE, seed=25, sv_every=15000.0, coverage=30.0, supp_frac=0.3, clip_pileups=2, contig_len=800_000, n_snp=1000), dict(), True, {}, {}),
    "short_reads_dense_mod": (dict(BASE, seed=26, sv_every=5000.0, len_median=3000.0, len_min=500, coverage=25.0), dict(mod_every=300.0), True, {}, {}),
    # clips on every second read + overlapping supplementary pieces: rows reached through the forward reach of clips, MOD rows behind the last SNP
    "clips_sparse_snps": (dict(BASE, seed=27, sv_every=15000.0, clip_every=2, supp_frac=0.5, supp_overlap_frac=1.0, n_snp=150), dict(mod_every=500.0), True, {}, {}),
    "window_threshold": (dict(BASE, seed=28, sv_every=15000.0, indel_var_frac=0.3, sub_rate=0.05, ins_rate=0.04, del_rate=0.04), dict(),
                         True, dict(phase_indel=1, connect_adjacent=20), dict(sv_window=3, sv_threshold=0.3)),
    "dense_everything": (dict(BASE, seed=29, sv_every=4000.0, n_snp=2000, coverage=8.0), dict(mod_every=200.0), True, dict(is_ont=0), {}),
    # more recorded rows per alignment than the kernel keeps in LDS (512): the second walk
    "mod_every_20": (dict(BASE, seed=30, contig_len=200_000, n_snp=200, coverage=6.0, len_median=40000.0), dict(mod_every=20.0, pair_frac=0.0, listed=1.0), False, {}, {}),
}


def build(name):
    kw, mod_kw, use_sv, pkw, xkw = CASES[name]
    s = Synth(**kw)
    lines = make_mod_lines(s, seed=kw["seed"], **mod_kw) if mod_kw is not None else []
    mpos, mrows = merge_mod_lines(lines)
    X = abi.ExtraVariants(s.sv_pos if use_sv else (), s.sv_len if use_sv else (), mpos, mrows, **xkw)
    V = abi.Variants(s.var_pos, s.var_ref, s.var_alt)
    R = abi.Reads.from_synth(s)
    return s, V, X, R, abi.default_params(**pkw)


@pytest.mark.parametrize("name", list(CASES))
def test_extra_rows_every_stage(name):
    s, V, X, R, P = build(name)
    want, wsv, wmod, d = lps_oracle.phase_x(P, V, X, s.ref, R, dump=True)
    assert (X.n_sv == 0 or (wsv.phase_set != 0).sum() > 0.5 * X.n_sv) and (X.n_mod == 0 or (wmod.phase_set != 0).sum() > 0.5 * X.n_mod)
    with hip.Context(0, P) as ctx:
        ctx.load_chromosome(V, s.ref, R)
        ctx.set_extra(X)
        for call in range(2):                      # the second call reuses the buffers of the first (and, with CNV intervals, waits for them)
            out = ctx.run_phase()
            util.assert_stages_equal(ctx, d, f"{name}, call {call}")
            util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, f"{name}: SNP rows")
            gsv, gmod = ctx.extra_result()
            util.assert_phase_equal(gsv.phase_set, gsv.gt, wsv.phase_set, wsv.gt, f"{name}: SV rows")
            util.assert_phase_equal(gmod.phase_set, gmod.gt, wmod.phase_set, wmod.gt, f"{name}: MOD rows")
        # without the extra rows the same ctx gives the plain result again
        ctx.set_extra(None)
        plain = ctx.run_phase()
        w0 = lps_oracle.phase(P, V, s.ref, R)[0]
        util.assert_phase_equal(plain.phase_set, plain.gt, w0.phase_set, w0.gt, f"{name}: SNP rows after the extra rows were dropped")


def test_extra_rows_as_bam_records_and_small_arenas():
    """The same through raw BAM records, from arenas that are too small for the merged rows (the library grows them and runs again)."""
    s, V, X, R, P = build("short_reads_dense_mod")
    want, wsv, wmod, d = lps_oracle.phase_x(P, V, X, s.ref, R, dump=True)
    B = abi.BamRecords.from_reads(R, seed=5)
    with hip.Context(0, P) as ctx:
        ctx.load_chromosome(V, s.ref, [B])
        ctx.set_extra(X)
        ctx._check(ctx.L.lps_debug_set_obs_capacity(ctx.h, int(d.c.n_obs * 0.8)), "lps_debug_set_obs_capacity")
        out = ctx.run_phase()
        util.assert_stages_equal(ctx, d, "BAM records + small arenas")
        gsv, gmod = ctx.extra_result()
        util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "SNP rows")
        util.assert_phase_equal(gsv.phase_set, gsv.gt, wsv.phase_set, wsv.gt, "SV rows")
        util.assert_phase_equal(gmod.phase_set, gmod.gt, wmod.phase_set, wmod.gt, "MOD rows")


def test_extra_rows_refused():
    """A position in two of the three tables, unsorted rows, unsorted read lists: refused (the reference would not terminate / misparse)."""
    s, V, X, R, P = build("sv_and_mod")
    with hip.Context(0, P) as ctx:
        ctx.load_chromosome(V, s.ref, R)
        for bad, msg in ((abi.ExtraVariants([int(s.var_pos[10])], [300], (), ()), b"more than one"),
                         (abi.ExtraVariants((), (), [int(s.var_pos[10])], [[(1, True, False)]]), b"more than one"),
                         (abi.ExtraVariants([5000], [300], [5000], [[(1, True, False)]]), b"more than one"),
                         (abi.ExtraVariants([9000, 5000], [300, 200], (), ()), b"strictly increasing"),
                         (abi.ExtraVariants([5000], [300], (), (), sv_threshold=1.5), b"svThreshold")):
            assert ctx.L.lps_set_extra_variants(ctx.h, bad.c) != 0
            assert msg in ctx.L.lps_last_error(ctx.h), ctx.L.lps_last_error(ctx.h)
        out = ctx.run_phase()                       # a refused table leaves none set
        w0 = lps_oracle.phase(P, V, s.ref, R)[0]
        util.assert_phase_equal(out.phase_set, out.gt, w0.phase_set, w0.gt, "after refused tables")
