"""Fixture matrix shared by make_golden.py (generator of the golden vectors) and the parity tests.

Each fixture = generator arguments (longphase-s_amd/lps/synth.py), the reference CLI flags used when the golden
output was produced, and the matching lps_params overrides.  Matrix follows SURVEY.md §8c.
"""
import hashlib
import numpy as np

SMALL = dict(contig_len=400_000, n_snp=500, coverage=15.0, n_threads=4)

PHASE_FIXTURES = {
    # name: (synth kwargs, reference CLI flags, lps_params overrides)
    "snp_ont": (dict(SMALL, seed=1), ["--ont"], {}),
    "snp_ont_seed2": (dict(SMALL, seed=2, coverage=30.0), ["--ont"], {}),
    "snp_pb": (dict(SMALL, seed=3), ["--pb"], dict(is_ont=0)),
    "indels": (dict(SMALL, seed=4, indel_var_frac=0.25), ["--ont", "--indels"], dict(phase_indel=1)),
    "indels_pb": (dict(SMALL, seed=5, indel_var_frac=0.4, tandem_frac=0.6), ["--pb", "--indels"], dict(is_ont=0, phase_indel=1)),
    "two_blocks": (dict(SMALL, seed=6, contig_len=900_000, n_snp=900, gap_start=300_000, gap_len=320_000), ["--ont"], {}),
    "lowq_heavy": (dict(SMALL, seed=7, lowq_frac=0.5), ["--ont"], {}),
    "supp_overlap": (dict(SMALL, seed=8, supp_frac=0.35, supp_overlap_frac=0.8), ["--ont"], {}),
    "cnv_pileup": (dict(SMALL, seed=9, contig_len=800_000, n_snp=1000, coverage=40.0, clip_pileups=2), ["--ont"], {}),
    "cnv_many": (dict(SMALL, seed=16, contig_len=2_000_000, n_snp=2400, coverage=45.0, clip_pileups=7, supp_frac=0.05), ["--ont"], {}),
    # 80 simulated break points -> 64 distinct CNV intervals (128 entries in the reference's cnvVec): beyond the 32 the first GPU version could hold
    "cnv_64": (dict(SMALL, seed=18, contig_len=5_000_000, n_snp=5000, coverage=35.0, clip_pileups=80, supp_frac=0.03), ["--ont"], {}),
    "sparse_cov": (dict(SMALL, seed=10, coverage=3.0), ["--ont"], {}),
    "dense_snps": (dict(SMALL, seed=11, n_snp=4000, snp_pair_frac=0.08, snp_in_hpoly_frac=0.3, hpoly_every=300.0), ["--ont"], {}),
    "params_a": (dict(SMALL, seed=12), ["--ont", "-a", "20", "-d", "50000", "-q", "20", "-p", "20", "-e", "0.3"],
                 dict(connect_adjacent=20, distance=50000, mapping_quality=20, base_quality=20, edge_weight=0.3)),
    "params_b": (dict(SMALL, seed=13, indel_var_frac=0.2), ["--ont", "--indels", "-m", "0.8", "-n", "0.9", "-1", "0.5", "-L", "0.05"],
                 dict(phase_indel=1, read_confidence=0.8, snp_confidence=0.9, edge_threshold=0.5, overlap_threshold=0.05)),
    "high_error": (dict(SMALL, seed=14, sub_rate=0.06, ins_rate=0.04, del_rate=0.04), ["--ont"], {}),
    "short_reads": (dict(SMALL, seed=15, len_median=3000.0, len_min=500, coverage=25.0), ["--ont"], {}),
    # half of the reads split into overlapping primary + supplementary pieces, dense SNPs, many low-quality bases: merged reads hold positions
    # twice, where the order std::sort leaves among equal positions reaches the fp32 edge sums
    "supp_light_dups": (dict(SMALL, seed=17, n_snp=1500, coverage=25.0, supp_frac=0.5, supp_overlap_frac=1.0, lowq_frac=0.4), ["--ont"], {}),
}

# `phase --sv-file --mod-file` fixtures: name -> (synth kwargs, make_mod_lines kwargs or None (no MOD file), use the generator's SVs,
# reference CLI flags, lps_params overrides, lps_extra_variants overrides).  The MOD read lists are drawn by numpy and therefore stored in the
# golden file next to the reference's results.
XB = dict(contig_len=400_000, n_snp=500, coverage=15.0, n_threads=4)
EXTRA_FIXTURES = {
    "sv_and_mod": (dict(XB, seed=21, sv_every=15000.0), dict(), True, ["--ont"], {}, {}),
    "sv_only": (dict(XB, seed=22, sv_every=15000.0), None, True, ["--ont"], {}, {}),
    "mod_only": (dict(XB, seed=23), dict(), False, ["--ont"], {}, {}),
    "pb_indels": (dict(XB, seed=24, sv_every=15000.0, indel_var_frac=0.3), dict(), True, ["--pb", "--indels"], dict(is_ont=0, phase_indel=1), {}),
    "supp": (dict(XB, seed=25, sv_every=15000.0, coverage=30.0, supp_frac=0.3, clip_pileups=2, contig_len=800_000, n_snp=1000), dict(), True, ["--ont"], {}, {}),
    # break-point clip pile-ups: CNV intervals, the mismatch filter looks at SV / MOD rows like at any other
    "cnv": (dict(XB, seed=16, sv_every=20000.0, contig_len=2_000_000, n_snp=2400, coverage=45.0, clip_pileups=7, supp_frac=0.05), dict(mod_every=4000.0), True, ["--ont"], {}, {}),
    "short_reads_dense_mod": (dict(XB, seed=26, sv_every=5000.0, len_median=3000.0, len_min=500, coverage=25.0), dict(mod_every=300.0), True, ["--ont"], {}, {}),
    # clips on every second read + overlapping supplementary pieces: rows reached through the forward reach of clips, MOD rows behind the last SNP
    "clips_sparse_snps": (dict(XB, seed=27, sv_every=15000.0, clip_every=2, supp_frac=0.5, supp_overlap_frac=1.0, n_snp=150), dict(mod_every=500.0), True, ["--ont"], {}, {}),
    "window_threshold": (dict(XB, seed=28, sv_every=15000.0, indel_var_frac=0.3, sub_rate=0.05, ins_rate=0.04, del_rate=0.04), dict(), True,
                         ["--ont", "--indels", "-a", "20", "--svWindow", "3", "--svThreshold", "0.3"], dict(phase_indel=1, connect_adjacent=20), dict(sv_window=3, sv_threshold=0.3)),
    "dense_everything": (dict(XB, seed=29, sv_every=4000.0, n_snp=2000, coverage=8.0), dict(mod_every=200.0), True, ["--pb"], dict(is_ont=0), {}),
    # more recorded rows per alignment than the kernel keeps in LDS (512): its second walk
    "mod_every_20": (dict(XB, seed=30, contig_len=200_000, n_snp=200, coverage=6.0, len_median=40000.0), dict(mod_every=20.0, pair_frac=0.0, listed=1.0), False, ["--ont"], {}, {}),
}

# haplotag fixtures: (phase fixture providing reads + the reference's own phased VCF, haplotag CLI flags, params overrides)
HAPLOTAG_FIXTURES = {
    "snp_ont": ("snp_ont", [], {}),
    "indels": ("indels", [], {}),
    "two_blocks": ("two_blocks", [], {}),
    "sparse_cov": ("sparse_cov", [], {}),
    "supp_tagged": ("supp_overlap", ["--tagSupplementary"], dict(tag_supplementary=1)),
    "dense_snps": ("dense_snps", [], {}),
    "strict": ("lowq_heavy", ["-q", "20", "-p", "0.8"], dict(mapping_quality=20, percentage_threshold=0.8)),
    "high_error": ("high_error", [], {}),
}

# end-to-end fixtures of the haplotag CLI (reads regenerated from the seed, stale HP/PS/PQ + other optional fields added by
# util.add_stale_tags, phased VCF written from the haplotag fixture's table): name -> haplotag fixture
CLI_HAPLOTAG_FIXTURES = ["snp_ont", "indels", "supp_tagged", "strict", "two_blocks"]

# multi-contig end-to-end fixture of the CLI: (contig, synth kwargs, has VCF records); the BAM also ends with unplaced reads
MULTI = dict(contig_len=150_000, n_snp=200, coverage=14.0, n_threads=2)
MULTI_CONTIG_FIXTURE = [("chrA", dict(MULTI, seed=41), True), ("chrB", dict(MULTI, seed=42, contig_len=90_000, n_snp=110, supp_frac=0.2), True),
                        ("chrEmpty", dict(MULTI, seed=43, contig_len=50_000, n_snp=60), False), ("chrC", dict(MULTI, seed=44, indel_var_frac=0.2), True)]

# end-to-end fixture of `phase --sv-file --mod-file`: three contigs in one BAM, the last one without SNP records; the SV / MOD VCFs (with the records
# the reference's readers drop, see make_golden.make_cli_extra) are committed under tests/golden/data next to the three VCFs the reference wrote
XC = dict(n_threads=2, coverage=14.0)
CLI_EXTRA_FIXTURE = [("chrA", dict(XC, seed=51, contig_len=120_000, n_snp=160, sv_every=8000.0), True),
                     ("chrB", dict(XC, seed=52, contig_len=90_000, n_snp=110, sv_every=6000.0, supp_frac=0.2), True),
                     ("chrC", dict(XC, seed=53, contig_len=50_000, n_snp=60, sv_every=6000.0), False)]

# tumor/normal fixtures for the somatic rows: (genome kwargs, normal reads kwargs, tumor reads kwargs, somatic_haplotag CLI, params)
TN_BASE = dict(contig_len=600_000, n_snp=700, n_threads=4, somatic_every=8000.0)
SOMATIC_FIXTURES = {
    "tn60": (dict(TN_BASE, seed=31), dict(coverage=25.0, read_seed=311, tumor_purity=0.0), dict(coverage=50.0, read_seed=312, tumor_purity=0.6), [], {}),
    "tn30_indel": (dict(TN_BASE, seed=32, indel_var_frac=0.2), dict(coverage=20.0, read_seed=321, tumor_purity=0.0),
                   dict(coverage=40.0, read_seed=322, tumor_purity=0.3), ["--tagSupplementary"], dict(tag_supplementary=1)),
    "tn90_blocks": (dict(TN_BASE, seed=33, contig_len=900_000, n_snp=900, gap_start=300_000, gap_len=320_000, somatic_every=5000.0),
                    dict(coverage=25.0, read_seed=331, tumor_purity=0.0), dict(coverage=45.0, read_seed=332, tumor_purity=0.9, supp_frac=0.2), ["-p", "0.7"], dict(percentage_threshold=0.7)),
}

# end-to-end fixtures of the somatic_haplotag CLI: name -> (tumor/normal fixture (or its own spec), --tumor-purity, extra CLI flags)
SOMATIC_CLI_ONLY = {}
SOMATIC_CLI_ONLY["tn_dense"] = (dict(TN_BASE, seed=34, somatic_every=1200.0), dict(coverage=25.0, read_seed=341, tumor_purity=0.0), dict(coverage=50.0, read_seed=342, tumor_purity=0.8), [], {})
ALL_SOMATIC = dict(SOMATIC_FIXTURES, **SOMATIC_CLI_ONLY)
CLI_SOMATIC_FIXTURES = {"tn60_p06": ("tn60", "0.6", []), "tn30_indel_p03": ("tn30_indel", "0.3", []), "tn90_blocks_p095": ("tn90_blocks", "0.95", []),
                        "tn_dense_p08": ("tn_dense", "0.8", []), "tn_dense_p015": ("tn_dense", "0.15", []), "tn60_nofilter": ("tn60", "0.6", ["--disableFilter"]),
                        "tn60_auto": ("tn60", "auto", []), "tn30_indel_auto": ("tn30_indel", "auto", []), "tn90_blocks_auto": ("tn90_blocks", "auto", []), "tn_dense_auto": ("tn_dense", "auto", [])}

# fixtures whose full inputs (FASTA/VCF/SAM) are committed as data files under tests/golden/data/
TINY = dict(contig_len=60_000, n_snp=120, coverage=12.0, n_threads=2)
DATA_FIXTURES = {
    "tiny_snp": (dict(TINY, seed=21), ["--ont"], {}),
    "tiny_indel": (dict(TINY, seed=22, indel_var_frac=0.3), ["--ont", "--indels"], dict(phase_indel=1)),
}


def input_digest(s):
    """sha256 over the generated inputs: detects generator drift between the machine that made the golden
    vectors and the one that replays them."""
    h = hashlib.sha256()
    for a in (s.var_pos, s.ref_start, s.flag, s.mapq, s.l_qseq, s.name_id, s.cigar_off, s.cigar, s.seq, s.qual):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(b"|".join(s.var_ref))
    h.update(b"|".join(s.var_alt))
    h.update(np.ascontiguousarray(s.ref).tobytes())
    return h.hexdigest()
