"""CPU tests: the oracle's restatement of `phase --sv-file --mod-file` (three-cursor get_snp src/phase/ParsingBam.cpp:1303-1634 with its SV
and MOD branches, node types 1 / 2, the SNP<->MOD edge threshold src/phase/PhasingGraph.cpp:197-202, readCorrection's SV / MOD cases) must
reproduce what the REAL reference binary wrote to <prefix>.vcf, <prefix>_SV.vcf and <prefix>_mod.vcf (tests/golden/make_golden.py --extra)."""
import numpy as np
import pytest

import fixtures
import lps_oracle
import util


@pytest.mark.parametrize("name", sorted(fixtures.EXTRA_FIXTURES))
def test_oracle_matches_reference_with_sv_and_mod_rows(name):
    s, V, X, R, P, g = util.make_extra_case(name)
    out, osv, omod, d = lps_oracle.phase_x(P, V, X, s.ref, R, dump=True)
    util.assert_phase_equal(out.phase_set, out.gt, g["phase_set"], g["gt"], name + ": SNP rows")
    util.assert_phase_equal(osv.phase_set, osv.gt, g["sv_ps"], g["sv_gt"], name + ": SV rows")
    util.assert_phase_equal(omod.phase_set, omod.gt, g["mod_ps"], g["mod_gt"], name + ": MOD rows")
    idx = util.INDEX["extra:" + name]
    assert ((out.phase_set != 0).sum(), (osv.phase_set != 0).sum(), (omod.phase_set != 0).sum()) == (idx["n_phased"], idx["n_sv_phased"], idx["n_mod_phased"])
    if name == "cnv":
        assert d.c.n_cnv > 0
    if name not in ("clips_sparse_snps", "mod_every_20"):      # these two reach MOD rows behind the last SNP on purpose (end() comparison of the reference)
        assert d.c.ub_hazard == 0


def test_oracle_refuses_rows_the_reference_would_spin_on():
    s, V, X, R, P, g = util.make_extra_case("sv_only")
    from lps import abi
    tie = abi.ExtraVariants([int(V.pos[40])], [300], (), ())
    with pytest.raises(RuntimeError):
        lps_oracle.phase_x(P, V, tie, s.ref, R)
