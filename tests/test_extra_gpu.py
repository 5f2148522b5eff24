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

pytestmark = pytest.mark.gpu

import fixtures


@pytest.mark.parametrize("name", sorted(fixtures.EXTRA_FIXTURES))
def test_extra_rows_every_stage(name):
    s, V, X, R, P, g = util.make_extra_case(name)
    want, wsv, wmod, d = lps_oracle.phase_x(P, V, X, s.ref, R, dump=True)
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
        # ... and what the reference binary wrote for the same input
        util.assert_phase_equal(out.phase_set, out.gt, g["phase_set"], g["gt"], f"{name}: SNP rows vs reference")
        util.assert_phase_equal(gsv.phase_set, gsv.gt, g["sv_ps"], g["sv_gt"], f"{name}: SV rows vs reference")
        util.assert_phase_equal(gmod.phase_set, gmod.gt, g["mod_ps"], g["mod_gt"], f"{name}: MOD rows vs reference")
        # without the extra rows the same ctx gives the plain result again
        ctx.set_extra(None)
        plain = ctx.run_phase()
        w0 = lps_oracle.phase(P, V, s.ref, R)[0]
        util.assert_phase_equal(plain.phase_set, plain.gt, w0.phase_set, w0.gt, f"{name}: SNP rows after the extra rows were dropped")


def test_extra_rows_as_bam_records_and_small_arenas():
    """The same through raw BAM records, from arenas that are too small for the merged rows (the library grows them and runs again)."""
    s, V, X, R, P, _ = util.make_extra_case("short_reads_dense_mod")
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
    s, V, X, R, P, _ = util.make_extra_case("sv_and_mod")
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
