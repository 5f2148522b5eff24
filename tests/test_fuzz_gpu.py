"""GPU fuzz parity: many small random contigs (low coverage, high error, dense/sparse SNPs, indels) through the HIP path
vs the CPU oracle.  Exercises ties / new blocks / skipped nodes in the vote scan, empty rows, tiny graphs."""
import numpy as np
import pytest

import lps_oracle
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu


def test_fuzz_small_contigs():
    rng = np.random.default_rng(7)
    ctx = {}
    n_cases = 40
    for case in range(n_cases):
        kw = dict(seed=1000 + case, contig_len=int(rng.integers(20_000, 200_000)), n_snp=int(rng.integers(5, 600)),
                  coverage=float(rng.choice([1.0, 2.0, 4.0, 8.0, 20.0])), len_median=float(rng.choice([2000.0, 8000.0, 15000.0])),
                  len_min=500, sub_rate=float(rng.choice([0.01, 0.08])), indel_var_frac=float(rng.choice([0.0, 0.3])),
                  lowq_frac=float(rng.choice([0.1, 0.6])), supp_frac=float(rng.choice([0.02, 0.3])), n_threads=2,
                  snp_pair_frac=float(rng.choice([0.01, 0.1])), snp_in_hpoly_frac=0.2, hpoly_every=500.0)
        over = dict(phase_indel=1) if kw["indel_var_frac"] > 0 else {}
        if case % 5 == 0:
            over.update(connect_adjacent=int(rng.integers(2, 63)), distance=int(rng.choice([2000, 300000])))
        s, V, R = util.make_case(kw)
        if V.n == 0 or R.n_reads == 0:
            continue
        P = abi.default_params(**over)
        ref_out, d = lps_oracle.phase(P, V, s.ref, R, dump=True)
        key = tuple(sorted(over.items()))
        if key not in ctx:
            ctx[key] = hip.Context(0, P)
        out = ctx[key].phase(V, s.ref, R)
        hp, blk = ctx[key].dump_votes()
        N = d.c.n_nodes
        assert np.array_equal(hp, d.node_hp[:N]) and np.array_equal(blk, d.node_block[:N]), f"case {case}: scan differs {kw} {over}"
        util.assert_phase_equal(out.phase_set, out.gt, ref_out.phase_set, ref_out.gt, f"case {case} {kw} {over}")
    for c in ctx.values():
        c.close()
