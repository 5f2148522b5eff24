"""End-to-end check of `longphase_amd somatic_haplotag`, with an explicit --tumor-purity and with the automatic estimator: the three
BAM passes run on the GPU (rows a20-a22), the caller's per-site statistics, filters and the purity estimator are restated on the host.  Against the reference binary on the
same files: the per-site filter log (every intermediate value and all six filter decisions) must be identical text, the number of flagged somatic
variants equal, and the inflated record stream of the tagged tumor BAM (HP:Z / PS:i / PQ:i) byte-identical."""
import hashlib
import json
import os
import subprocess

import pytest

import fixtures
import util

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(HERE, "..", "longphase-s_amd", "cli", "longphase_amd")
pytestmark = pytest.mark.gpu


# both BAMs through zlib on the host (the default below 256 MiB), or inflated on the GPU where they stay: records pushed from the resident streams, the
# tagged BAM spliced and deflated there (lps_somatic_write_bgzf) - for a third of the fixtures, to keep the suite short
_KEYS = sorted(fixtures.CLI_SOMATIC_FIXTURES)
@pytest.mark.parametrize("key,inflate", [(k, "host") for k in _KEYS] + [(k, "gpu") for k in _KEYS[::3]])
def test_cli_somatic_matches_reference(key, inflate, tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", f"cli_somatic_{key}.json")))
    name, purity, extra = fixtures.CLI_SOMATIC_FIXTURES[key]
    d = str(tmp_path)
    digests, _ = util.make_somatic_inputs(d, name)
    assert list(digests) == gold["digests"], "generator drift"
    util.write_bam(d + "/normal.sam", d + "/normal.bam"); util.write_bam(d + "/tumor.sam", d + "/tumor.bam", block=40000)
    phased = os.path.join(HERE, "golden", "data", f"somatic_{name}.normal_phased.vcf")
    r = subprocess.run([CLI, "somatic_haplotag", "-s", phased, "-b", "normal.bam", "--tumor-snv-file", "tumor.vcf", "--tumor-bam-file", "tumor.bam", "-r", "ref.fa", "-t", "4",
                        "-o", "som", "--somatic-calling-log", "--output-somatic-vcf"] + (["--tumor-purity", purity] if purity != "auto" else []) + gold["cli"] + ["--" + inflate + "-inflate"], cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, LPS_CLI_DEBUG="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    print(r.stderr[-700:])
    if purity == "auto":                                              # the estimator's report: every count, the box-plot statistics and the purity itself
        assert open(d + "/som_purity.out").read() == gold["purity_out"]
    got_log = open(d + "/som_somatic_filter.log").read().splitlines(); want_log = gold["filter_log"].splitlines()
    assert len(got_log) == len(want_log)
    for g, w in zip(got_log, want_log):
        assert g == w
    assert int([l for l in r.stderr.splitlines() if l.startswith("somatic variant count(Flag)")][0].split(":")[1]) == gold["flag_count"]
    sc = [l for l in open(d + "/som_sc.vcf").read().split("\n") if not l.startswith("##commandline=") and not l.startswith("##longphase_s_version=")]
    assert sum(1 for l in sc if l and not l.startswith("#") and l.split("\t")[6] == "PASS") == gold["sc_vcf_pass"]
    assert hashlib.sha256("\n".join(sc).encode()).hexdigest() == gold["sc_vcf_sha256"]
    text, refs, recs = util.bam_sections(d + "/som.bam")
    assert [l for l in text.split("\n") if l and not l.startswith("@PG")] == gold["header_without_pg"]
    got = util.bam_record_tags(recs)
    want = [(q, f, p, [tuple(t) for t in tg]) for q, f, p, tg in gold["tags"]]
    assert len(got) == gold["n_records"]
    for g, w in zip(got, want):
        assert g == w
    assert hashlib.sha256(recs).hexdigest() == gold["records_sha256"]


def test_cli_somatic_argument_errors(tmp_path):
    r = subprocess.run([CLI, "somatic_haplotag", "-s", "a.vcf", "-b", "a.bam", "--tumor-snv-file", "t.vcf"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "missing arguments" in r.stderr
    r = subprocess.run([CLI, "somatic_haplotag", "-s", "a.vcf", "-b", "a.bam", "--tumor-snv-file", "t.vcf", "--tumor-bam-file", "t.bam", "-r", "r.fa", "--cram"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "not supported" in r.stderr


def _merge_contigs(d, parts):
    """three single-contig tumor/normal fixtures -> one three-contig input set in d (text merges: headers first, records in contig order)"""
    def cat(name, is_header):
        head, body = [], []
        for sub, _ in parts:
            for ln in open(os.path.join(d, sub, name)):
                (head if is_header(ln) else body).append(ln)
        seen, uniq = set(), []
        for h in head:
            if h not in seen:
                seen.add(h); uniq.append(h)
        return uniq, body
    with open(d + "/ref.fa", "w") as f:
        for sub, _ in parts:
            f.write(open(os.path.join(d, sub, "ref.fa")).read())
    for sam in ("normal.sam", "tumor.sam"):
        head, body = cat(sam, lambda l: l.startswith("@"))
        hd = [h for h in head if h.startswith("@HD")][:1] + [h for h in head if h.startswith("@SQ")] + [h for h in head if not h.startswith(("@HD", "@SQ"))]
        open(os.path.join(d, sam), "w").write("".join(hd + body))
    for vcf in ("tumor.vcf", "normal_phased.vcf"):
        head, body = cat(vcf, lambda l: l.startswith("#"))
        cols = [h for h in head if h.startswith("#CHROM")][:1]
        open(os.path.join(d, vcf), "w").write("".join([h for h in head if h.startswith("##")] + cols + body))


def test_cli_somatic_gpus_equals_single_worker(tmp_path):
    """somatic_haplotag --gpus 3 (three contexts on the one GPU of the box, contigs dealt onto them, purity estimated over all contigs, outputs merged in
    contig order) must write exactly what the single-worker run writes: filter log, purity report, somatic VCF and the tagged BAM's record stream."""
    d = str(tmp_path)
    parts = [("a", "chrA"), ("b", "chrB"), ("c", "chrC")]
    for (sub, chrom), name in zip(parts, ("tn60", "tn30_indel", "tn_dense")):
        os.makedirs(os.path.join(d, sub))
        util.make_somatic_inputs(os.path.join(d, sub), name, chrom=chrom)
        phased = open(os.path.join(HERE, "golden", "data", f"somatic_{name}.normal_phased.vcf")).read().replace("chrS", chrom)
        open(os.path.join(d, sub, "normal_phased.vcf"), "w").write(phased)
    _merge_contigs(d, parts)
    util.write_bam(d + "/normal.sam", d + "/normal.bam"); util.write_bam(d + "/tumor.sam", d + "/tumor.bam", block=40000)
    outs = {}
    for tag, extra in (("one", []), ("three", ["--gpus", "3"]), ("gpuin", ["--gpu-inflate"]), ("gpuin_hostz", ["--gpu-inflate", "--host-deflate"])):
        r = subprocess.run([CLI, "somatic_haplotag", "-s", "normal_phased.vcf", "-b", "normal.bam", "--tumor-snv-file", "tumor.vcf", "--tumor-bam-file", "tumor.bam", "-r", "ref.fa",
                            "-t", "4", "-o", tag, "--somatic-calling-log", "--output-somatic-vcf", "--tagSupplementary"] + extra, cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert ("3 workers" in r.stderr) == (tag == "three")
        text, refs, recs = util.bam_sections(os.path.join(d, tag + ".bam"))
        sc = [l for l in open(os.path.join(d, tag + "_sc.vcf")).read().split("\n") if not l.startswith("##commandline=")]
        outs[tag] = (hashlib.sha256(recs).hexdigest(), len(recs), open(os.path.join(d, tag + "_somatic_filter.log")).read(), open(os.path.join(d, tag + "_purity.out")).read(), sc,
                     [l for l in r.stderr.splitlines() if l.startswith("somatic variant count(Flag)")])
    assert outs["one"][1] > 1_000_000 and len(outs["one"][2].splitlines()) > 100
    assert outs["one"] == outs["three"]
    assert outs["one"] == outs["gpuin"]                                 # both streams resident on the GPU, three contigs taken from them, the GPU's tag writer
    assert outs["one"] == outs["gpuin_hostz"]                           # inflated on the GPU, copied back once, three contigs cut out of the copy
    # both BAMs indexed: the pair is walked in contig groups (one group of the tumor BAM + the same contigs of the normal BAM resident at a time, loaded
    # once per phase - the purity estimation walks all groups first), here one contig per group, all three in one group, and the indexes ignored
    util.write_bai(d + "/normal.bam"); util.write_bai(d + "/tumor.bam")
    # ... and the three groups dealt onto three workers (`--gpus 3`: each with two contexts and its own view of the two files, all on the one GPU of the box)
    for tag, extra, n_groups in (("grp1", ["--gpu-inflate", "--group-bytes", "1"], 3), ("grp_all", ["--gpu-inflate"], 1), ("noidx", ["--gpu-inflate", "--no-index"], 0),
                                 ("grp1_3w", ["--gpu-inflate", "--group-bytes", "1", "--gpus", "3"], 3)):
        r = subprocess.run([CLI, "somatic_haplotag", "-s", "normal_phased.vcf", "-b", "normal.bam", "--tumor-snv-file", "tumor.vcf", "--tumor-bam-file", "tumor.bam", "-r", "ref.fa",
                            "-t", "4", "-o", tag, "--somatic-calling-log", "--output-somatic-vcf", "--tagSupplementary"] + extra, cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert (f"contig groups: {n_groups} " in r.stderr) == (n_groups > 0), r.stderr[-600:]
        assert ("3 workers" in r.stderr) == (tag == "grp1_3w")
        text, refs, recs = util.bam_sections(os.path.join(d, tag + ".bam"))
        sc = [l for l in open(os.path.join(d, tag + "_sc.vcf")).read().split("\n") if not l.startswith("##commandline=")]
        got = (hashlib.sha256(recs).hexdigest(), len(recs), open(os.path.join(d, tag + "_somatic_filter.log")).read(), open(os.path.join(d, tag + "_purity.out")).read(), sc,
               [l for l in r.stderr.splitlines() if l.startswith("somatic variant count(Flag)")])
        assert got == outs["one"], tag
