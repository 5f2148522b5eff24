"""Fuzz parity of the SV / MOD rows (csrc/lps_extra.hip) on HAND-MADE alignments: random CIGARs with every operation code the reference accepts -
long soft / hard clips and insertions whose length reaches forward over rows they do not consume, deletions and skips with SNPs inside (the
`break` of the reference's inner loop), pads, =/X - random SV and modcall rows between random SNPs, shared read names, both strands.  The GPU
serves the rows by a closed form; the oracle walks the reference's three cursors literally.  Observations (union indices, alleles, qualities)
and the phased result of all three tables must agree."""
import numpy as np
import pytest

import lps_oracle
import util
from lps import abi, hip

pytestmark = pytest.mark.gpu

M, I, D, N, S, H, P, EQ, X = range(9)


def make_case(seed):
    g = np.random.default_rng(seed)
    L = int(g.integers(30_000, 120_000))
    ref = g.choice(np.frombuffer(b"ACGT", np.uint8), L)
    for _ in range(L // 400):                                        # homopolymer runs (filterSNP / the deletion branch look at them)
        p = int(g.integers(10, L - 20)); ref[p:p + int(g.integers(3, 9))] = ref[p]
    n_reads = int(g.integers(60, 400))
    sv_like = [100, 200, 300]
    reads = []
    for r in range(n_reads):
        start = int(g.integers(0, L - 8000))
        ops = []
        if g.random() < 0.4:
            ops.append((S if g.random() < 0.7 else H, int(g.choice([3, 8, 40, 700, 5000]))))
        pos = start
        for _ in range(int(g.integers(3, 70))):
            ops.append((int(g.choice([M, M, M, EQ, X])), int(g.integers(1, 250))))
            pos += ops[-1][1]
            u = g.random()
            if u < 0.25: ops.append((I, int(g.choice([1, 2, 5, 30] + sv_like))))
            elif u < 0.5: ops.append((D, int(g.choice([1, 2, 7, 60] + sv_like)))); pos += ops[-1][1]
            elif u < 0.56: ops.append((N, int(g.integers(1, 1500)))); pos += ops[-1][1]
            elif u < 0.6: ops.append((P, int(g.integers(1, 40))))
            if pos > L - 3000:
                break
        if ops[-1][0] not in (M, EQ, X):
            ops.append((M, int(g.integers(1, 50))))
        if g.random() < 0.4:
            ops.append((S if g.random() < 0.7 else H, int(g.choice([2, 9, 300, 4000]))))
        merged = []
        for o, l in ops:                                             # neighbouring equal codes are legal in BAM but keep the CIGARs canonical
            if merged and merged[-1][0] == o: merged[-1] = (o, merged[-1][1] + l)
            else: merged.append((o, l))
        lq = sum(l for o, l in merged if o in (M, I, S, EQ, X))
        flag = (16 if g.random() < 0.5 else 0) | (0x800 if g.random() < 0.1 else 0) | (0x100 if g.random() < 0.02 else 0)
        reads.append((start, merged, lq, flag, int(g.choice([0, 20, 60, 60, 60])), int(g.integers(0, max(2, n_reads * 3 // 4)))))
    reads.sort(key=lambda t: t[0])
    cig = np.array([(l << 4) | o for _, ops, _, _, _, _ in reads for o, l in ops], np.uint32)
    cig_off = np.concatenate([[0], np.cumsum([len(t[1]) for t in reads])]).astype(np.uint64)
    lqs = np.array([t[2] for t in reads], np.int32)
    seq_off = np.concatenate([[0], np.cumsum((lqs + 1) // 2)]).astype(np.uint64); qual_off = np.concatenate([[0], np.cumsum(lqs)]).astype(np.uint64)
    nib = g.choice(np.array([1, 2, 4, 8], np.uint8), int(seq_off[-1]) * 2)
    seq = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8)
    qual = g.integers(0, 60, int(qual_off[-1])).astype(np.uint8)
    R = abi.Reads(ref_start=np.array([t[0] for t in reads], np.int32), flag=np.array([t[3] for t in reads], np.uint16), mapq=np.array([t[4] for t in reads], np.uint8),
                  l_qseq=lqs, name_id=np.array([t[5] for t in reads], np.uint32), cigar_off=cig_off, cigar=cig, seq_off=seq_off, seq=seq, qual_off=qual_off, qual=qual)
    # three tables without a common position
    n_snp = int(g.integers(20, 500))
    allp = g.choice(np.arange(5, L - 50), size=min(L - 60, n_snp + int(g.integers(0, 120)) + int(g.integers(0, 300))), replace=False)
    snp = np.sort(allp[:n_snp]); rest = allp[n_snp:]
    n_sv = int(g.integers(0, min(len(rest), 120) + 1)); sv = np.sort(rest[:n_sv]); mod = np.sort(rest[n_sv:])
    alt = np.array([g.choice([b for b in b"ACGT" if b != ref[p]]) for p in snp], np.uint8)
    V = abi.Variants(snp, [bytes([ref[p]]) for p in snp], [bytes([a]) for a in alt])
    sv_len = np.array([int(g.choice([99, 199, 299, 120, 450])) * int(g.choice([1, -1])) for _ in sv], np.int32)
    names = np.unique(R.name_id)
    rows = []
    for _ in mod:
        k = int(g.integers(0, min(len(names), 40) + 1))
        rows.append([(int(nm), bool(g.random() < 0.5), bool(g.random() < 0.5)) for nm in g.choice(names, size=k, replace=False)])
    X_ = abi.ExtraVariants(sv, sv_len, mod, rows, sv_window=int(g.choice([1, 3, 20])), sv_threshold=float(g.choice([0.05, 0.1, 0.5])))
    P_ = abi.default_params(is_ont=int(g.random() < 0.5), mapping_quality=int(g.choice([1, 30])))
    return ref, V, X_, R, P_


@pytest.mark.parametrize("block", range(4))
def test_fuzz_extra_rows_on_random_cigars(block):
    n_obs = n_extra = 0
    for seed in range(500 + 12 * block, 500 + 12 * (block + 1)):
        ref, V, X, R, P = make_case(seed)
        want, wsv, wmod, d = lps_oracle.phase_x(P, V, X, ref, R, dump=True)
        with hip.Context(0, P) as ctx:
            ctx.load_chromosome(V, ref, R)
            ctx.set_extra(X)
            out = ctx.run_phase()
            util.assert_stages_equal(ctx, d, f"seed {seed}")
            gsv, gmod = ctx.extra_result()
            # the same alignments without the two extra tables: the plain path on CIGARs no generator makes (N, P, H, =, X, clips inside)
            ctx.set_extra(None)
            plain = ctx.run_phase()
            w0, d0 = lps_oracle.phase(P, V, ref, R, dump=True)
            util.assert_stages_equal(ctx, d0, f"seed {seed}, SNP table alone")
            util.assert_phase_equal(plain.phase_set, plain.gt, w0.phase_set, w0.gt, f"seed {seed}: SNP table alone")
            # ... and the haplotag scorer on the table that phase run produced
            idx = np.nonzero(plain.phase_set != 0)[0]
            if idx.size:
                VT = abi.Variants(V.pos[idx], [V.ref_str[i] for i in idx], [V.alt_str[i] for i in idx], hp1_is_alt=plain.gt[idx], phase_set=plain.phase_set[idx])
                ctx.set_table(VT, ref)
                tag = ctx.run_haplotag()
                wt = lps_oracle.haplotag(P, VT, ref, R)
                for k in ("status", "hp1", "hp2", "ps_min", "hp", "pq", "ps"):
                    assert np.array_equal(getattr(tag, k), getattr(wt, k)), f"seed {seed}: haplotag {k}"
        util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, f"seed {seed}: SNP rows")
        util.assert_phase_equal(gsv.phase_set, gsv.gt, wsv.phase_set, wsv.gt, f"seed {seed}: SV rows")
        util.assert_phase_equal(gmod.phase_set, gmod.gt, wmod.phase_set, wmod.gt, f"seed {seed}: MOD rows")
        q = d.obs_quality[:d.c.n_obs]
        n_obs += int(d.c.n_obs); n_extra += int(((q == -1) | (q == -2) | (q == -3)).sum())
    assert n_extra > 500 and n_obs > 5 * n_extra // 10, (n_obs, n_extra)
