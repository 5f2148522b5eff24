"""End-to-end check of `longphase_amd haplotag`: the inflated record stream of its output BAM must be
byte-identical to the one the reference binary wrote for the same inputs (tests/golden/cli_haplotag_*.json holds its sha256 and
per-record tags): every record is written, stale HP/PS/PQ fields are stripped from scored records only, new tags are appended as
HP:i PS:i PQ:i (src/haplotag/HaplotagProcess.cpp:337-361), everything else is copied untouched.  Headers may differ in @PG only."""
import hashlib
import json
import os
import subprocess

import pytest

import fixtures
import util
from lps.synth import Synth

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(HERE, "..", "longphase-s_amd", "cli", "longphase_amd")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,inflate", [(n, "gpu") for n in fixtures.CLI_HAPLOTAG_FIXTURES] + [("supp_tagged", "host"), ("indels", "host"), ("strict", "hostdeflate"), ("two_blocks", "hostdeflate")])
def test_cli_haplotag_output_bam_matches_reference(name, inflate, tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", f"cli_haplotag_{name}.json")))
    src, tag_cli, over = fixtures.HAPLOTAG_FIXTURES[name]
    kw, _, _ = fixtures.PHASE_FIXTURES[src]
    s = Synth(**kw)
    assert fixtures.input_digest(s) == gold["digest"], "generator drift"
    V, _, _, _ = util.load_golden_haplotag(name)
    d = str(tmp_path)
    s.write_fasta(d + "/ref.fa"); s.write_sam(d + "/plain.sam")
    util.add_stale_tags(d + "/plain.sam", d + "/reads.sam")
    util.write_table_vcf(d + "/table.vcf", V, "chrS", kw["contig_len"])
    util.write_bam(d + "/reads.sam", d + "/reads.bam", block=20000)
    s.close()
    r = subprocess.run([CLI, "haplotag", "-s", "table.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", "4", "-o", "tagged"] + tag_cli + (["--host-inflate"] if inflate == "host" else ["--gpu-inflate", "--host-deflate"] if inflate == "hostdeflate" else ["--gpu-inflate"]),
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    text, refs, recs = util.bam_sections(d + "/tagged.bam")
    in_text, in_refs, _ = util.bam_sections(d + "/reads.bam")
    assert refs == in_refs
    assert [l for l in text.split("\n") if l and not l.startswith("@PG")] == gold["header_without_pg"]
    assert sum(1 for l in text.split("\n") if l.startswith("@PG")) == 1
    got = util.bam_record_tags(recs)
    want = [(q, f, p, [tuple(t) for t in tg]) for q, f, p, tg in gold["tags"]]
    assert len(got) == gold["n_records"]
    for g, w in zip(got, want):
        assert g == w
    assert len(recs) == gold["record_bytes"]
    assert hashlib.sha256(recs).hexdigest() == gold["records_sha256"]
    assert ("gpu tag splice+deflate" in r.stderr) == (inflate == "gpu")
    # the output is a valid BGZF file: ends with the 28-byte EOF block, every block <= 64 KiB
    raw = open(d + "/tagged.bam", "rb").read()
    assert raw[-28:] == bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


@pytest.mark.parametrize("mode", ["gpu", "host", "gpu_indexed_2workers"])
def test_cli_haplotag_with_phased_sv_and_mod_files(mode, tmp_path):
    """`haplotag --sv-file --mod-file` (judgeSVHap, src/haplotag/HaplotagStrategy.cpp:220-226; the lists are read by
    src/haplotag/HaplotagVcfParser.cpp:403-468): the three phased VCFs the reference's `phase --sv-file --mod-file` wrote go in, the record stream
    of the tagged BAM must be the reference's - 232 of its 303 records carry other tags than without the two files."""
    import gzip
    gold = json.load(open(os.path.join(HERE, "golden", "cli_haplotag_extra.json")))
    kw = fixtures.EXTRA_FIXTURES["sv_and_mod"][0]
    s = Synth(**kw)
    assert fixtures.input_digest(s) == gold["digest"], "generator drift"
    d = str(tmp_path)
    s.write_fasta(d + "/ref.fa"); s.write_sam(d + "/plain.sam"); s.close()
    util.add_stale_tags(d + "/plain.sam", d + "/reads.sam")
    util.write_bam(d + "/reads.sam", d + "/reads.bam", block=20000)
    for fn in ("out.vcf", "out_SV.vcf", "out_mod.vcf"):
        open(os.path.join(d, fn), "w").write(gzip.open(os.path.join(HERE, "golden", "data", "cli_haplotag_extra." + fn + ".gz"), "rt").read())
    flags = ["--host-inflate"] if mode == "host" else ["--gpu-inflate"]
    if mode.startswith("gpu_indexed"):
        util.write_bai(d + "/reads.bam"); flags += ["--gpus", "2"]
    r = subprocess.run([CLI, "haplotag", "-s", "out.vcf", "-b", "reads.bam", "-r", "ref.fa", "-t", "4", "-o", "tagged", "--sv-file", "out_SV.vcf", "--mod-file", "out_mod.vcf"] + flags,
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    text, refs, recs = util.bam_sections(d + "/tagged.bam")
    got = util.bam_record_tags(recs)
    want = [(q, f, p, [tuple(t) for t in tg]) for q, f, p, tg in gold["tags"]]
    assert len(got) == gold["n_records"]
    for g, w in zip(got, want):
        assert g == w
    assert hashlib.sha256(recs).hexdigest() == gold["records_sha256"]
    assert gold["records_changed_by_the_votes"] > 100
