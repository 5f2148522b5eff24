"""Parity at the sizes BASELINE.json's configs name: the HIP path through the C-ABI against the CPU oracle, bit for bit,
on chr20-30x (configs[1]) and on one 50x contig of 160 Mb (the per-contig unit of configs[2]/[3]) - observations, clips, nodes, the fp32
edge matrix as uint32, vote-scan hp/block, PS/GT, and the haplotag counts/tags on the table the phase run produced.  The inputs come from
the GPU generator (tools/lps_synth_gpu.hip): the host generator needs minutes for these sizes.  Scale-only paths covered here: arena
overflow re-run, ~5 000 speculative scan segments, the sort-key packing with 19/20-bit fields, 64 observation arenas in use."""
import ctypes as C

import numpy as np
import pytest

import lps_oracle
import util
from lps import abi, hip
from lps.synth_gpu import SynthGpu

pytestmark = pytest.mark.gpu

CHR20_30X = dict(seed=101, contig_len=64_444_167, n_snp=60_000, coverage=30.0)
CONTIG_50X = dict(seed=201, contig_len=160_000_000, n_snp=206_000, coverage=50.0)          # ~ GRCh38 chr7 at the WGS density of 4 M SNPs / 3.1 Gb


def check_haplotag(ctx, P, V, ph, ref, R, what):
    idx = np.nonzero(ph.phase_set != 0)[0]
    VT = abi.Variants.from_snps(V.pos[idx], V.ref0[idx], V.alt0[idx], hp1_is_alt=ph.gt[idx], phase_set=ph.phase_set[idx])
    ctx.set_table(VT, ref)
    out = ctx.run_haplotag()
    want = lps_oracle.haplotag(P, VT, ref, R)
    for k in ("status", "hp1", "hp2", "ps_min", "hp", "pq", "ps"):
        assert np.array_equal(getattr(out, k), getattr(want, k)), f"{what}: haplotag {k} differs"
    assert np.array_equal(np.minimum(out.n_ps, 2), np.minimum(want.n_ps, 2))
    assert (out.hp != 0).sum() > 0.5 * R.n_reads


def test_chr20_30x_every_stage_host_push():
    """configs[1] through lps_push_reads (host arrays), every stage dump compared."""
    g = SynthGpu(0, **CHR20_30X)
    h = g.to_host(); g.close()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0); R = abi.Reads.from_synth(h)
    assert R.n_reads > 90_000 and V.n > 59_000
    P = abi.default_params()
    want, d = lps_oracle.phase(P, V, h.ref, R, dump=True)
    assert d.c.ub_hazard == 0
    with hip.Context(0, P) as ctx:
        out = ctx.phase(V, h.ref, R)
        util.assert_stages_equal(ctx, d, "chr20_30x")
        util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "chr20_30x vs oracle")
        assert (out.phase_set != 0).sum() > 0.95 * V.n
        # the same run from arenas that are far too small: the library must notice, grow them and come to the same result
        ctx._check(ctx.L.lps_debug_set_obs_capacity(ctx.h, 70_000), "lps_debug_set_obs_capacity")
        out2 = ctx.run_phase()
        util.assert_stages_equal(ctx, d, "chr20_30x after the arena overflow")
        util.assert_phase_equal(out2.phase_set, out2.gt, want.phase_set, want.gt, "chr20_30x after the arena overflow")
        check_haplotag(ctx, P, V, out, h.ref, R, "chr20_30x")


@pytest.mark.parametrize("every", [2, 13, 97, 400])
def test_chr20_30x_vote_scan_with_unmatched_boundaries(every, monkeypatch):
    """~950 speculative scan segments of which every `every`-th boundary is declared unmatched: k_scan_stitch composes up to the break, replays
    that segment from the true state, composes on - votes, blocks and the result stay those of the oracle; and a handful of breaks must not cost
    a serial pass over all segments."""
    g = SynthGpu(0, **CHR20_30X)
    h = g.to_host(); g.close()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0); R = abi.Reads.from_synth(h)
    P = abi.default_params()
    want, d = lps_oracle.phase(P, V, h.ref, R, dump=True)
    with hip.Context(0, P) as ctx:
        monkeypatch.setenv("LPS_SCAN_FORCE_REPLAY", str(every))
        out = ctx.phase(V, h.ref, R)
        tm = ctx.timings()
        assert tm["n_scan_replayed"] >= tm["n_scan_segments"] // every - 1
        hp, blk = ctx.dump_votes()
        N = d.c.n_nodes
        assert np.array_equal(hp, d.node_hp[:N]) and np.array_equal(blk, d.node_block[:N]), "vote scan differs"
        util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, f"every {every}")
        ctx.L.lps_set_stage_timing(ctx.h, 2); ctx.run_phase(); tm = ctx.timings()
        print(f"every {every}: {tm['n_scan_replayed']} of {tm['n_scan_segments']} segments replayed, vote_scan {tm['stages']['vote_scan']:.3f} ms")
        if every >= 97:
            assert tm["stages"]["vote_scan"] < 0.6


def test_contig_160mb_50x_every_stage_device_push():
    """One 50x contig of the whole-genome configs through lps_push_reads_device (arrays generated in HBM), every stage dump compared."""
    g = SynthGpu(0, **CONTIG_50X)
    h = g.to_host()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0); R = abi.Reads.from_synth(h)
    assert h.contig_len >= 150_000_000 and R.n_reads > 350_000
    P = abi.default_params()
    with hip.Context(0, P) as ctx:
        ctx.load_chromosome_device(V, h.ref, g.device_batch(), g.n_reads)
        g.close()
        out = ctx.run_phase()
        tm = ctx.timings()
        want, d = lps_oracle.phase(P, V, h.ref, R, dump=True)
        assert d.c.ub_hazard == 0
        util.assert_stages_equal(ctx, d, "160 Mb 50x")
        util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "160 Mb 50x vs oracle")
        assert tm["n_scan_segments"] > 3000 and tm["n_obs"] > 8_000_000
        check_haplotag(ctx, P, V, out, h.ref, R, "160 Mb 50x")


def test_chr20_30x_with_sv_and_mod_rows():
    """configs[1] with ~30 000 modcall rows and ~400 SV rows co-phased (`--sv-file --mod-file`): every stage in indices of the union table, the
    three results; then the same from arenas that cannot hold the merged rows."""
    from lps.synth import make_extras_fast
    g = SynthGpu(0, **CHR20_30X)
    h = g.to_host(); g.close()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0); R = abi.Reads.from_synth(h)
    X = abi.extra_from_arrays(*make_extras_fast(h, seed=7))
    assert X.n_mod > 25_000 and X.n_sv > 300
    P = abi.default_params()
    want, wsv, wmod, d = lps_oracle.phase_x(P, V, X, h.ref, R, dump=True)
    assert (wmod.phase_set != 0).sum() > 0.8 * X.n_mod             # (no generated read carries the SV rows: they are served, called REF, and stay unphased)
    with hip.Context(0, P) as ctx:
        ctx.load_chromosome(V, h.ref, R)
        ctx.set_extra(X)
        for call in range(2):
            if call == 1:
                ctx._check(ctx.L.lps_debug_set_obs_capacity(ctx.h, int(d.c.n_obs) // 2), "lps_debug_set_obs_capacity")
            out = ctx.run_phase()
            util.assert_stages_equal(ctx, d, f"chr20_30x + SV/MOD rows, call {call}")
            gsv, gmod = ctx.extra_result()
            util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "SNP rows")
            util.assert_phase_equal(gsv.phase_set, gsv.gt, wsv.phase_set, wsv.gt, "SV rows")
            util.assert_phase_equal(gmod.phase_set, gmod.gt, wmod.phase_set, wmod.gt, "MOD rows")
        ctx.L.lps_set_stage_timing(ctx.h, 2)
        ctx.run_phase()
        tm = ctx.timings()
        with_rows = tm["stages"]["extract"]
        ctx.set_extra(None); ctx.run_phase()
        print(f"extract stage: {ctx.timings()['stages']['extract']:.3f} ms without, {with_rows:.3f} ms with {X.n_mod} MOD + {X.n_sv} SV rows served and merged; whole step {tm['ms_total']:.3f} ms")


def test_chr20_30x_with_clip_pile_ups_cnv_filter_active():
    """configs[1] with 50 simulated break points: clip pile-ups give CNV intervals (replayed on the host), the late stages run with the CNV
    mismatch filter; the second call takes the path that waits for the intervals instead of guessing "none"."""
    g = SynthGpu(0, clip_pileups=50, **CHR20_30X)
    h = g.to_host(); g.close()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0); R = abi.Reads.from_synth(h)
    P = abi.default_params()
    want, d = lps_oracle.phase(P, V, h.ref, R, dump=True)
    assert d.c.ub_hazard == 0 and d.c.n_cnv >= 40, d.c.n_cnv
    with hip.Context(0, P) as ctx:
        for call in range(2):
            out = ctx.phase(V, h.ref, R) if call == 0 else ctx.run_phase()
            util.assert_stages_equal(ctx, d, f"chr20_30x pile-ups, call {call}")
            util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, f"chr20_30x pile-ups, call {call}")


def test_ultra_long_read_cg_tag_cigar_and_row_longer_than_65536():
    """One read of 70 000 CIGAR operations over 77 000 dense variants: as a BAM record its CIGAR sits in a CG:B,I field (more than 65 535
    operations, htslib expands it behind sam_itr_multi_next), its row of observations is longer than 2^16 (the old sort-key field), and the
    extraction sends it through k_extract_redo's direct-to-memory path."""
    g = SynthGpu(0, seed=77, contig_len=400_000, n_snp=79_000, coverage=3.0, hpoly_every=400)
    h = g.to_host(); g.close()
    R0 = abi.Reads.from_synth(h)
    # the long read: (10M 1I) x 35 000 from position 1 000, bases copied from the reference (ALT at every 3rd variant), inserted bases 'A'
    start, n_blk = 1000, 35_000
    code = {65: 1, 67: 2, 71: 4, 84: 8}
    ref = h.ref.copy()
    alt_at = {int(p): int(a) for p, a in zip(h.var_pos[::3], h.var_alt0[::3])}
    span = ref[start:start + 10 * n_blk].copy()
    for p, a in alt_at.items():
        if start <= p < start + 10 * n_blk:
            span[p - start] = a
    q = np.empty(11 * n_blk, np.uint8)
    q.reshape(n_blk, 11)[:, :10] = span.reshape(n_blk, 10); q.reshape(n_blk, 11)[:, 10] = 65
    nib = np.vectorize(code.get)(q).astype(np.uint8)
    seq = (nib[0::2] << 4) | np.append(nib[1::2], 0)[:(q.size + 1) // 2]
    qual = np.full(q.size, 30, np.uint8)
    cig = np.tile(np.array([(10 << 4) | 0, (1 << 4) | 1], np.uint32), n_blk)
    at = int(np.searchsorted(R0.ref_start, start, side="right"))
    def ins(a, v): return np.concatenate([a[:at], np.asarray(v, a.dtype), a[at:]])
    def ins_rows(off, data, row):
        o = off.astype(np.int64); cut = int(o[at])
        return np.concatenate([o[:at + 1], o[at:] + len(row)]).astype(np.uint64), np.concatenate([data[:cut], row, data[cut:]])
    co, cg = ins_rows(R0.cigar_off, R0.cigar, cig); so, sq = ins_rows(R0.seq_off, R0.seq, seq); qo, ql = ins_rows(R0.qual_off, R0.qual, qual)
    R = abi.Reads(ref_start=ins(R0.ref_start, [start]), flag=ins(R0.flag, [0]), mapq=ins(R0.mapq, [60]), l_qseq=ins(R0.l_qseq, [q.size]),
                  name_id=ins(R0.name_id, [int(R0.name_id.max()) + 7]), cigar_off=co, cigar=cg, seq_off=so, seq=sq, qual_off=qo, qual=ql)
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0)
    P = abi.default_params()
    want, d = lps_oracle.phase(P, V, h.ref, R, dump=True)
    assert d.obs_count.max() > 65536
    B = abi.BamRecords.from_reads(R, seed=3)
    for reads, what in ((R, "decoded arrays"), (B, "BAM records with a CG field")):
        with hip.Context(0, P) as ctx:
            out = ctx.phase(V, h.ref, [reads])
            util.assert_stages_equal(ctx, d, "ultra-long read, " + what)
            util.assert_phase_equal(out.phase_set, out.gt, want.phase_set, want.gt, "ultra-long read, " + what)


def test_device_push_rejects_bad_operands():
    """lps_push_reads_device runs the operand checks of lps_push_reads as a kernel."""
    g = SynthGpu(0, seed=5, contig_len=400_000, n_snp=400, coverage=8.0)
    h = g.to_host()
    V = abi.Variants.from_snps(h.var_pos, h.var_ref0, h.var_alt0)
    with hip.Context(0, abi.default_params()) as ctx:
        ctx.load_chromosome_device(V, h.ref, g.device_batch(), g.n_reads)           # fine as generated
        b = g.device_batch()
        b.qual_off, b.seq_off = b.seq_off, b.qual_off                                # qual rows now half as long as l_qseq needs
        ctx.L.lps_begin_chromosome(ctx.h)
        assert ctx.L.lps_push_reads_device(ctx.h, C.byref(b)) != 0
        assert b"l_qseq" in ctx.L.lps_last_error(ctx.h)
    g.close()


def test_tumor_normal_pair_50x_25x_every_somatic_pass():
    """BASELINE configs[4] at one contig: a tumor / normal pair (50x / 25x, SNP + indel variants, 60 % purity) on a 16 Mb contig - the normal
    sample is phased on the GPU, the merged table (phased germline rows + the tumor-only somatic rows) goes through the three per-read passes
    of somatic_haplotag (normal extraction a20, tumor extraction a21, tagging a22), each compared with the oracle: every per-site counter,
    per-read count, list entry and tag."""
    from lps.synth import Synth
    genome = dict(seed=61, contig_len=16_000_000, n_snp=16_000, indel_var_frac=0.15, somatic_every=6000.0, n_threads=8)
    N = Synth(**dict(genome, coverage=25.0, read_seed=611, tumor_purity=0.0))
    T = Synth(**dict(genome, coverage=50.0, read_seed=612, tumor_purity=0.6))
    RN, RT = abi.Reads.from_synth(N), abi.Reads.from_synth(T)
    assert RT.n_reads > 35_000 and RN.n_reads > 17_000 and N.n_somatic > 1500
    P = abi.default_params(phase_indel=1)
    V0 = abi.Variants(N.var_pos, N.var_ref, N.var_alt)
    with hip.Context(0, P) as ctx:
        ph = ctx.phase(V0, N.ref, RN)
        util.assert_phase_equal(ph.phase_set, ph.gt, *(lambda o: (o.phase_set, o.gt))(lps_oracle.phase(P, V0, N.ref, RN)[0]), "normal sample")
        # merged table: phased germline rows (role 0) + somatic SNVs of the tumor VCF (role 1, derived from the haplotype they sit on)
        keep = np.nonzero(ph.phase_set != 0)[0]
        rows = [(int(N.var_pos[i]), N.var_ref[i], N.var_alt[i], int(ph.gt[i]), int(ph.phase_set[i]), 0, 0) for i in keep]
        rows += [(int(p), bytes([r]), bytes([a]), 0, 0, 1, int(h) + 1) for p, r, a, h in zip(N.som_pos, N.som_ref, N.som_alt, N.som_hap)]
        rows.sort()
        assert len(set(r[0] for r in rows)) == len(rows)
        kind = [1 if len(r[1]) == 1 and len(r[2]) == 1 else (2 if len(r[1]) == 1 else 3) for r in rows]
        V = abi.Variants([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], hp1_is_alt=[r[3] for r in rows], phase_set=[r[4] for r in rows],
                         somatic_role=[r[5] for r in rows], derive_hp=[r[6] for r in rows], tumor_kind=kind)
        # a20: normal sample at the tumor-VCF positions
        want = lps_oracle.somatic_extract_normal(P, V, N.ref, RN)
        out = ctx.somatic_extract_normal(V, N.ref, RN)
        assert np.array_equal(out.read_hp, want.read_hp) and np.array_equal(out.counters, want.counters), "normal extraction"
        # a21: tumor sample
        want = lps_oracle.somatic_extract_tumor(P, V, T.ref, RT)
        out = ctx.somatic_extract_tumor(V, T.ref, RT)
        for k in ("status", "hp1", "hp2", "hp3", "hp", "ps_min", "end_pos", "read_len", "has_site"):
            assert np.array_equal(getattr(out, k), getattr(want, k)), "tumor extraction: " + k
        assert np.array_equal(out.site, want.site), "tumor extraction: per-site counters"
        assert out.c.n_pairs == want.c.n_pairs and out.c.n_windows == want.c.n_windows and out.c.n_pairs > 50_000
        for a, b in zip(out.pairs(), want.pairs()):
            assert np.array_equal(a, b), "tumor extraction: (site, read, base HP) pairs"
        for a, b in zip(out.windows(), want.windows()):
            assert np.array_equal(a, b), "tumor extraction: difference windows"
        # a22: tagging
        want = lps_oracle.somatic_tag(P, V, RT)
        out = ctx.somatic_tag(V, T.ref, RT)
        for k in ("status", "hp1", "hp2", "hp3", "derive_h1", "derive_h2", "ps_min", "hp", "pq", "ps"):
            assert np.array_equal(getattr(out, k), getattr(want, k)), "tagging: " + k
        assert (out.hp >= 5).sum() > 1000 and (out.hp == 1).sum() > 5000


def test_tumor_normal_pair_160mb_three_passes_as_the_command_line_runs_them():
    """BASELINE configs[4] at scale: tumor 50x / normal 25x at 60 % purity on a 160 Mb contig (8 Gbases + 4 Gbases of alignments), the merged table as
    `somatic_haplotag` builds it - the normal sample's phased rows plus the tumor VCF's somatic rows (role 2 for the extraction passes, role 1 with
    their derived haplotype for the tagging pass).  bench.py's tumor / normal leg runs the three passes from the resident alignments and compares every
    per-site counter, per-read count, (site, read, base HP) pair, difference window and tag with the oracle."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    r = bench.somatic_leg(0, 160.0, 1, min(16, ncpu), check=True)
    assert r["parity"] == {"normal_extract": True, "tumor_extract": True, "tag": True}, r["parity"]
    assert r["tumor_alignments"] > 380_000 and r["normal_alignments"] > 190_000 and r["somatic_rows"] > 5_000
    assert r["pairs"] > 100_000 and r["windows"] > 100_000 and r["tagged_somatic_reads"] > 10_000 and r["tagged_germline_reads"] > 200_000
    print({k: r[k] for k in ("value", "pass_ms", "kernel_ms", "pairs", "windows", "oracle_s", "generation_s")})
